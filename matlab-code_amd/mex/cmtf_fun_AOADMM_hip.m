function [G,out] = cmtf_fun_AOADMM_hip(Z,Znorm_const,G,fh,gh,lscalar,uscalar,options)
% Drop-in for cmtf_fun_AOADMM (same signature, functions/cmtf_fun_AOADMM.m:1) that runs the
% AO-ADMM outer loop on an MI355X through aoadmm_mex / libaoadmm_hip.so.
%
% Use: in functions/cmtf_AOADMM.m line 193 replace
%        [Fac,out] = cmtf_fun_AOADMM(Z,Znorm_const, G,fh,gh,lscalar,uscalar,options);
%      by
%        [Fac,out] = cmtf_fun_AOADMM_hip(Z,Znorm_const, G,fh,gh,lscalar,uscalar,options);
% Nothing else changes: Z, the 'init' struct, init_options and options keep their fields; optional
% engine settings live in options.hip (device = 0 or devices = [0 1 ... 7] for several GPUs from this one
% MATLAB process, precision = 'f64' | 'f32', par2_slab_sharding, no_permuted_copy).
%
% Function handles cannot cross to the GPU, so Z.prox_operators / Z.reg_func (cmtf_AOADMM.m:30-32) are
% dropped and the MEX gateway re-reads the constraint descriptors Z.constraints{m}. Models the device
% path does not cover ('custom' constraints, KL/IS/beta losses, sptensor data) raise cmtf:hip:unsupported, which is caught here and
% handed to the original MATLAB implementation, so every example script keeps running.
% Znorm_const, fh, gh, lscalar, uscalar are only needed by that fallback.

    Zs = Z;
    if isfield(Zs,'prox_operators'), Zs = rmfield(Zs,'prox_operators'); end
    if isfield(Zs,'reg_func'),       Zs = rmfield(Zs,'reg_func');       end
    for p = 1:numel(Zs.object)           % Tensor Toolbox objects -> plain double arrays
        if isa(Zs.object{p},'tensor')
            Zs.object{p} = double(Zs.object{p});
        end
    end
    if isfield(Zs,'miss')                % masks (sptensor / logical, cmtf_AOADMM.m:88-119) -> dense uint8, 1 = observed
        for p = 1:numel(Zs.miss)
            if isempty(Zs.miss{p}), continue; end
            if iscell(Zs.miss{p})
                Zs.miss{p} = cellfun(@(m) uint8(m ~= 0), Zs.miss{p}, 'UniformOutput', false);
            else
                Zs.miss{p} = uint8(double(full(Zs.miss{p})) ~= 0);
            end
        end
    end
    try
        tstart = tic;
        if any(strcmp(options.Display,{'iter','final'}))   % header, cmtf_fun_AOADMM.m:44-51
            if isfield(Zs,'miss') && ~isempty(Zs.miss) && any(~cellfun(@isempty,Zs.miss))
                fprintf(1,' Iter  f total      f tensors      f couplings    f constraints    f PAR2 couplings  f_rel_miss\n');
            else
                fprintf(1,' Iter  f total      f tensors      f couplings    f constraints    f PAR2 couplings\n');
            end
            fprintf(1,'------ ------------ -------------  -------------- ---------------- ----------------\n');
        end
        [G,out] = aoadmm_mex(Zs, G, options);   % with Display = 'iter' the gateway prints a row every DisplayIters iterations
        if any(strcmp(options.Display,{'iter','final'}))   % final row, cmtf_fun_AOADMM.m:496-503
            it = out.OuterIterations;
            ft = out.func_val_conv(it+1); fc = out.func_coupl_conv(it+1);
            fz = out.func_constr_conv(it+1); fp = out.func_PAR2_coupl(it+1);
            if isfield(out,'func_rel_missing')
                fprintf(1,'%6d %12f %12f %12f %12f %12f %12f\n', it, ft+fc+fz+fp, ft, fc, fz, fp, out.func_rel_missing(it+1));
            else
                fprintf(1,'%6d %12f %12f %12f %12f %12f\n', it, ft+fc+fz+fp, ft, fc, fz, fp);
            end
        end
        out.wall_time_hip = toc(tstart);
    catch err
        if strcmp(err.identifier,'cmtf:hip:unsupported')
            warning('cmtf:hip:fallback','%s -- running the MATLAB implementation instead.',err.message);
            [G,out] = cmtf_fun_AOADMM(Z,Znorm_const,G,fh,gh,lscalar,uscalar,options);
        else
            rethrow(err);
        end
    end
end
