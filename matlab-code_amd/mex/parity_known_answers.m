% parity_known_answers.m -- for whoever has MATLAB + Tensor Toolbox + the Proximity Operator Repository + TV_Condat_v2:
% runs the ORIGINAL cmtf_AOADMM on the four script-shaped known-answer cases of tests/golden/known_answers.py
% (models of example scripts 1, 13, 14 -- noise-free -- and 10 -- TV --, same data and the same init struct as the
% Python/HIP side) and prints Fit / factor-match scores and, with the HIP gateway built, the relative Frobenius gap of
% every factor after a fixed number of iterations.  Input: the .mat files written by
%     python tests/golden/make_golden.py --export-mat DIR
% The development container has no MATLAB: this script has never been executed there.
function parity_known_answers(dir_mat)
cases = {'script1','script13','script14','script10','script12'};
for c = 1:numel(cases)
    S = load(fullfile(dir_mat, [cases{c} '.mat']));
    [Z, G0, options] = build_case(cases{c}, S);
    [Zhat, Fac, ~, out] = cmtf_AOADMM(Z, 'alg_options', options, 'init', G0);            % original path, the script's tolerances
    fprintf('%s: %d outer iterations, f_tensors %.3e, f_couplings %.3e\n', cases{c}, out.OuterIterations, out.f_tensors, out.f_couplings);
    fixed = options; fixed.MaxOuterIters = 40;
    for f = {'AbsFuncTol','OuterRelTol','innerRelPrTol_coupl','innerRelPrTol_constr','innerRelDualTol_coupl','innerRelDualTol_constr'}
        fixed.(f{1}) = 0;
    end
    [~, FacRef] = cmtf_AOADMM(Z, 'alg_options', fixed, 'init', G0);
    if exist('cmtf_fun_AOADMM_hip', 'file')
        Zp = Z; [Zp.prox_operators, Zp.reg_func] = constraints_to_prox(Z.constrained_modes, Z.constraints, Z.size);
        nrm = cell(numel(Z.object), 1);
        for p = 1:numel(Z.object)
            if iscell(Z.object{p}), nrm{p} = sum(cellfun(@(x) norm(x, 'fro')^2, Z.object{p})); else, nrm{p} = norm(Z.object{p})^2; end
        end
        P = numel(Z.object);
        FacHip = cmtf_fun_AOADMM_hip(Zp, nrm, G0, cell(P,1), cell(P,1), cell(P,1), cell(P,1), fixed);
        for m = 1:numel(FacRef.fac)
            if iscell(FacRef.fac{m})
                g = max(cellfun(@(a, b) norm(a - b, 'fro') / norm(b, 'fro'), FacHip.fac{m}, FacRef.fac{m}));
            else
                g = norm(FacHip.fac{m} - FacRef.fac{m}, 'fro') / norm(FacRef.fac{m}, 'fro');
            end
            fprintf('  mode %d: relative Frobenius gap HIP vs MATLAB %.3e (target <= 1e-8)\n', m, g);
        end
    end
end
end

function [Z, G0, options] = build_case(name, S)
% the struct Z of the script the case is shaped after (see tests/golden/known_answers.py for the line references)
nn = {'non-negativity'};
options = S.options;
switch name
    case 'script1'
        K = size(S.truth_C, 1);
        Z.model = {'CP','PAR2'}; Z.modes = {[1 2 3],[4 5 6]}; Z.size = {20,30,40,20,30*ones(1,K),K};
        Z.coupling.lin_coupled_modes = [1 0 0 1 0 0]; Z.coupling.coupling_type = 0; Z.coupling.coupl_trafo_matrices = cell(6,1);
        Z.constrained_modes = [1 0 0 1 1 1]; Z.constraints = {nn,[],[],nn,nn,nn}; Z.weights = [1/2 1/2];
    case 'script13'
        Z.model = {'CP','CP'}; Z.modes = {[1 2 3],[4 5 6]}; Z.size = {50,30,40,100,70,80};
        H = cell(6,1); H{1} = eye(50); H{4} = zeros(50,100); for i = 1:50, H{4}(i, 2*i-1) = 1; end
        H2 = cell(6,1); H2{1} = eye(4); H2{4} = [eye(3); 0 0 0];
        Z.coupling.lin_coupled_modes = [1 0 0 1 0 0]; Z.coupling.coupling_type = 5;
        Z.coupling.coupl_trafo_matrices = H; Z.coupling.coupl_trafo_matrices2 = H2;
        Z.constrained_modes = [1 0 0 1 1 1]; Z.constraints = {nn,[],[],nn,nn,nn}; Z.weights = [1/2 1/2];
    case 'script14'
        K = size(S.truth_C, 1);
        Z.model = {'CP','PAR2'}; Z.modes = {[1 2 3],[4 5 6]}; Z.size = {20,30,40,20,30*ones(1,K),K};
        H = cell(6,1); H{1} = eye(20); H{6} = zeros(20,40); for i = 1:20, H{6}(i, 2*i-1) = 1; end
        Z.coupling.lin_coupled_modes = [1 0 0 0 0 1]; Z.coupling.coupling_type = 1; Z.coupling.coupl_trafo_matrices = H;
        Z.constrained_modes = [1 1 1 1 0 1]; Z.constraints = {nn,nn,nn,nn,[],nn}; Z.weights = [1/2 1/2];
    case 'script12'
        K = size(S.truth_C, 1);
        Z.model = {'CP','PAR2'}; Z.modes = {[1 2 3],[4 5 6]}; Z.size = {20,30,40,20,25*ones(1,K),K};
        Z.coupling.lin_coupled_modes = [1 0 0 1 0 0]; Z.coupling.coupling_type = 0; Z.coupling.coupl_trafo_matrices = cell(6,1);
        Z.constrained_modes = [0 0 0 0 0 0]; Z.constraints = cell(6,1); Z.weights = [1/2 1/2];
        Z.miss{1} = sptensor(tensor(double(S.miss_1)));                                   % example_script12_CP_PAR2_EM.m:108-113
        Z.miss{2} = arrayfun(@(k) logical(squeeze(S.miss_2(k,:,:))), 1:K, 'UniformOutput', false)';
    case 'script10'
        Z.model = {'CP'}; Z.modes = {[1 2 3]}; Z.size = {60,50,70};
        Z.coupling.lin_coupled_modes = [0 0 0]; Z.coupling.coupling_type = []; Z.coupling.coupl_trafo_matrices = cell(3,1);
        Z.constrained_modes = [1 1 1]; Z.constraints = {{'TV regularization',0.001},{'l2-ball',1},{'l2-ball',1}}; Z.weights = 1;
end
P = numel(Z.model);
Z.loss_function = repmat({'Frobenius'}, 1, P); Z.loss_function_param = cell(1, P);
for p = 1:P
    X = S.(sprintf('object_%d', p));
    if strcmp(Z.model{p}, 'PAR2')                     % stored as K x I x J_k
        Z.object{p} = arrayfun(@(k) squeeze(X(k,:,:)), 1:size(X,1), 'UniformOutput', false)';
    else
        Z.object{p} = tensor(X);
    end
end
% the init struct: arrays named init_<field>_<index>, cell-valued ones stacked along the first dimension
nm = numel(Z.size);
G0.fac = cell(nm,1); G0.constraint_fac = cell(nm,1); G0.constraint_dual_fac = cell(nm,1); G0.coupling_dual_fac = cell(nm,1);
G0.coupling_fac = cell(max(Z.coupling.lin_coupled_modes),1);
fn = fieldnames(S);
for i = 1:numel(fn)
    tok = regexp(fn{i}, '^init_(fac|constraint_fac|constraint_dual_fac|coupling_dual_fac|coupling_fac|DeltaB|P|mu_DeltaB)_(\d+)$', 'tokens', 'once');
    if isempty(tok), continue; end
    v = S.(fn{i}); idx = str2double(tok{2}) + 1;
    if isfield(S, [fn{i} '__cell']), v = arrayfun(@(k) squeeze(v(k,:,:)), 1:size(v,1), 'UniformOutput', false)'; end
    G0.(tok{1}){idx} = v;
end
end
