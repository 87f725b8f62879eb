function U = cmtf_nvecs_hip(Z,n,r)
% Drop-in for cmtf_nvecs (functions/cmtf_nvecs.m:1) for dense data: the I_n x I_n Gram matrix Y = A*A' of the
% mode-n unfolding (cmtf_nvecs.m:40-56) is computed on the MI355X through aoadmm_mex('unfold_gram',...), the
% r leading eigenvectors are taken with eigs exactly as in the reference (:58).
% Use: in functions/init_coupled_AOADMM_CMTF.m line 52 call cmtf_nvecs_hip instead of cmtf_nvecs.
    P = length(Z.object);
    for p = 1:P
        i = find(Z.modes{p} == n);
        if isempty(i), continue; end
        if isa(Z.object{p},'sptensor')
            U = cmtf_nvecs(Z,n,r);          % sparse data stay on the MATLAB path
            return
        end
        Y = aoadmm_mex('unfold_gram', double(Z.object{p}), i(1));
        [U,~] = eigs(Y, r, 'LM');
        return
    end
    error('cmtf:hip:usage','mode %d belongs to no data set', n);
end
