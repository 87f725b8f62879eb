% parity_dump_reference.m -- for whoever has MATLAB + Tensor Toolbox + the Proximity Operator Repository:
% runs the ORIGINAL cmtf_AOADMM and the HIP path on the same model / same init struct with all
% tolerances 0 (fixed iteration counts, SURVEY 8c) and prints the relative Frobenius gap per factor.
% The development container has no MATLAB, so this script has never been executed there; the CPU
% oracle (oracle/aoadmm.py) stands in for the left-hand side in tests/.
rng(4);
sz = {40,50,60}; R = 3;
A = cellfun(@(n) rand(n,R), sz, 'UniformOutput', false);
X = full(ktensor(A)); X = X + 0.05*norm(X)/sqrt(prod(cell2mat(sz)))*tensor(randn(size(X))); X = X/norm(X);
Z.loss_function = {'Frobenius'}; Z.loss_function_param = {[]}; Z.model = {'CP'}; Z.modes = {[1 2 3]}; Z.size = sz;
Z.coupling.lin_coupled_modes = [0 0 0]; Z.coupling.coupling_type = []; Z.coupling.coupl_trafo_matrices = cell(3,1);
Z.constrained_modes = [1 1 1]; Z.constraints = {{'non-negativity'},{'non-negativity'},{'non-negativity'}};
Z.weights = 1; Z.object{1} = X;
init_options.lambdas_init = {[1 1 1]}; init_options.nvecs = 0; init_options.normalize = 1;
init_options.distr = {@(x,y) rand(x,y), @(x,y) rand(x,y), @(x,y) rand(x,y)};
G0 = init_coupled_AOADMM_CMTF(Z,'init_options',init_options);
options = struct('Display','no','DisplayIters',10,'MaxOuterIters',20,'MaxInnerIters',5,'AbsFuncTol',0, ...
    'OuterRelTol',0,'innerRelPrTol_coupl',0,'innerRelPrTol_constr',0,'innerRelDualTol_coupl',0, ...
    'innerRelDualTol_constr',0,'bsum',0,'eps_log',1e-10);
[~,FacRef] = cmtf_AOADMM(Z,'alg_options',options,'init',G0,'init_options',init_options);   % original path
Zp = Z; [Zp.prox_operators, Zp.reg_func] = constraints_to_prox(Z.constrained_modes, Z.constraints, Z.size);
[FacHip,~] = cmtf_fun_AOADMM_hip(Zp, {norm(X)^2}, G0, {[]},{[]},{[]},{[]}, options);          % HIP path
for m = 1:3
    fprintf('mode %d: relative Frobenius gap %.3e (target <= 1e-8)\n', m, ...
        norm(FacHip.fac{m}-FacRef.fac{m},'fro')/norm(FacRef.fac{m},'fro'));
end
