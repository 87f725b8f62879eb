"""Shared test helpers: synthetic models shaped like the reference's example scripts."""
import numpy as np

from oracle.tensor_ops import full_ktensor


def rel_fro(a, b):
    a = np.asarray(a); b = np.asarray(b)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def options(**kw):
    """Option struct of example_script1_CP_PAR2_nonneg.m:110-123; tolerances 0 = fixed work."""
    o = dict(Display='no', DisplayIters=10, MaxOuterIters=10, MaxInnerIters=5, AbsFuncTol=0.0, OuterRelTol=0.0,
             innerRelPrTol_coupl=0.0, innerRelPrTol_constr=0.0, innerRelDualTol_coupl=0.0,
             innerRelDualTol_constr=0.0, bsum=0, eps_log=1e-10)
    o.update(kw)
    return o


def cp_data(dims, R, rng, noise=0.05, nonneg=True):
    """X = [[A1..AN]] + noise, normalised to ||X|| = 1 (create_coupled_data.m:158-162, example_script1:91-92)."""
    A = [rng.random((n, R)) if nonneg else rng.standard_normal((n, R)) for n in dims]
    X = full_ktensor(A)
    N = rng.standard_normal(X.shape)
    X = X + noise * np.linalg.norm(X) / np.linalg.norm(N) * N
    return X / np.linalg.norm(X), A


def cp_model(dims, R, rng, constraints, noise=0.05, weight=1.0):
    """Single uncoupled CP block with one constraint cell per mode (None = unconstrained)."""
    X, A = cp_data(dims, R, rng, noise)
    n = len(dims)
    Z = dict(loss_function=['Frobenius'], model=['CP'], modes=[list(range(1, n + 1))], size=list(dims),
             coupling=dict(lin_coupled_modes=[0] * n, coupling_type=[], coupl_trafo_matrices=[None] * n),
             constrained_modes=[0 if c is None else 1 for c in constraints], constraints=list(constraints),
             weights=[weight], object=[X])
    io = dict(lambdas_init=[[1] * R], nvecs=0, distr=[(lambda a, b: rng.random((a, b)))] * n, normalize=1)
    return Z, io, A


def script3_model(rng, noise=0.05, rows=50, R=4):
    """example_script3_matrix_CP_partialcoupling_nonneg.m:23-68: CP 50x30x40 R=4 + matrix 50x70 R=3,
    modes 1 and 4 coupled with type 4 (C = Delta*H), H1 = eye(4), H4 = [eye(3); 0 0 0].  `rows`: length of the coupled
    mode; `R`: rank of the tensor (the matrix has one column less)."""
    D = rng.random((rows, R))
    A = [D, rng.standard_normal((30, R)), rng.standard_normal((40, R))]
    M = [D[:, :R - 1], rng.random((70, R - 1))]
    X1 = full_ktensor(A)
    X2 = M[0] @ M[1].T
    for X in (X1, X2):
        N = rng.standard_normal(X.shape)
        X += noise * np.linalg.norm(X) / np.linalg.norm(N) * N
    X1 /= np.linalg.norm(X1)
    X2 /= np.linalg.norm(X2)
    H = [None] * 5
    H[0] = np.eye(R)
    H[3] = np.vstack([np.eye(R - 1), np.zeros((1, R - 1))])
    Z = dict(loss_function=['Frobenius'] * 2, model=['CP', 'CP'], modes=[[1, 2, 3], [4, 5]], size=[rows, 30, 40, rows, 70],
             coupling=dict(lin_coupled_modes=[1, 0, 0, 1, 0], coupling_type=[4], coupl_trafo_matrices=H),
             constrained_modes=[1, 0, 0, 1, 1],
             constraints=[('non-negativity',), None, None, ('non-negativity',), ('non-negative l2-sphere', 1)],
             weights=[0.5, 0.5], object=[X1, X2])
    distr = [lambda a, b: rng.random((a, b)), lambda a, b: rng.standard_normal((a, b)),
             lambda a, b: rng.standard_normal((a, b)), lambda a, b: rng.random((a, b)), lambda a, b: rng.random((a, b))]
    io = dict(lambdas_init=[[1] * R, [1] * (R - 1)], nvecs=0, distr=distr, normalize=1)
    return Z, io


def cp_cp_exact_model(rng, noise=0.05, rows=24, R=3):
    """Two CP tensors sharing their first factor exactly (coupling type 0), non-negative first modes."""
    D = rng.random((rows, R))
    A = [D, rng.standard_normal((18, R)), rng.random((20, R))]
    B = [D, rng.random((16, R)), rng.standard_normal((14, R))]
    X1 = full_ktensor(A)
    X2 = full_ktensor(B)
    for X in (X1, X2):
        N = rng.standard_normal(X.shape)
        X += noise * np.linalg.norm(X) / np.linalg.norm(N) * N
    X1 /= np.linalg.norm(X1)
    X2 /= np.linalg.norm(X2)
    Z = dict(loss_function=['Frobenius'] * 2, model=['CP', 'CP'], modes=[[1, 2, 3], [4, 5, 6]],
             size=[rows, 18, 20, rows, 16, 14],
             coupling=dict(lin_coupled_modes=[1, 0, 0, 1, 0, 0], coupling_type=[0], coupl_trafo_matrices=[None] * 6),
             constrained_modes=[1, 0, 1, 1, 1, 0],
             constraints=[('non-negativity',), None, ('non-negativity',), ('non-negativity',), ('box', 0.0, 2.0), None],
             weights=[0.5, 0.5], object=[X1, X2])
    distr = [lambda a, b: rng.random((a, b))] * 6
    io = dict(lambdas_init=[[1] * R, [1] * R], nvecs=0, distr=distr, normalize=1)
    return Z, io


def par2_slabs(I, Jk, R, rng, noise=0.0, nonneg_C=True):
    """PARAFAC2 data X_k = A diag(C_k) B_k' with B_k = P_k*DeltaB, P_k orthonormal
    (create_irregularPARAFAC2_coupled_data.m shape family), normalised to sum ||X_k||^2 = 1."""
    K = len(Jk)
    A = rng.standard_normal((I, R))
    C = rng.random((K, R)) + 0.1 if nonneg_C else rng.standard_normal((K, R))
    DB = rng.standard_normal((R, R))
    X = []
    for k in range(K):
        Q, _ = np.linalg.qr(rng.standard_normal((Jk[k], R)))
        Xk = A @ np.diag(C[k]) @ (Q @ DB).T
        if noise > 0:
            N = rng.standard_normal(Xk.shape)
            Xk = Xk + noise * np.linalg.norm(Xk) / np.linalg.norm(N) * N
        X.append(Xk)
    nrm = np.sqrt(sum(np.linalg.norm(x) ** 2 for x in X))
    return [x / nrm for x in X], A


def script4_model(rng, K=12, noise=0.2, constraints_B=None, R=3):
    """example_script4_irregularPAR2.m:18-51: PARAFAC2 I=40, ragged J_k from 61..120, R=3, C non-negative."""
    I = 40
    Jk = [61 + (7 * k) % 60 for k in range(K)]
    X, _ = par2_slabs(I, Jk, R, rng, noise)
    Z = dict(loss_function=['Frobenius'], model=['PAR2'], modes=[[1, 2, 3]], size=[I, Jk, K],
             coupling=dict(lin_coupled_modes=[0, 0, 0], coupling_type=[], coupl_trafo_matrices=[None] * 3),
             constrained_modes=[0, 1 if constraints_B else 0, 1], constraints=[None, constraints_B, ('non-negativity',)],
             weights=[1.0], object=[X])
    distr = [lambda a, b: rng.standard_normal((a, b)), lambda a, b: rng.standard_normal((a, b)),
             lambda a, b: rng.random((a, b)) + 0.1]
    io = dict(lambdas_init=[[1] * R], nvecs=0, distr=distr, normalize=1)
    return Z, io


def script1_model(rng, dims=(20, 30, 40), K=20, Jk=30, noise=0.0):
    """example_script1_CP_PAR2_nonneg.m:21-44: CP tensor + PARAFAC2 (I equal, K slabs) sharing their first
    factor exactly (modes 1 and 4, coupling type 0); non-negativity on modes 1,2,3,4,6."""
    R = 3
    I = dims[0]
    A = rng.random((I, R))
    X1 = full_ktensor([A, rng.random((dims[1], R)), rng.random((dims[2], R))])
    if noise > 0:
        N = rng.standard_normal(X1.shape)
        X1 = X1 + noise * np.linalg.norm(X1) / np.linalg.norm(N) * N
    X1 /= np.linalg.norm(X1)
    C = rng.random((K, R)) + 0.1
    DB = rng.standard_normal((R, R))
    Xk = []
    for k in range(K):
        Q, _ = np.linalg.qr(rng.standard_normal((Jk, R)))
        Xk.append(A @ np.diag(C[k]) @ (Q @ DB).T)
    nrm = np.sqrt(sum(np.linalg.norm(x) ** 2 for x in Xk))
    Xk = [x / nrm for x in Xk]
    Z = dict(loss_function=['Frobenius'] * 2, model=['CP', 'PAR2'], modes=[[1, 2, 3], [4, 5, 6]],
             size=[dims[0], dims[1], dims[2], I, [Jk] * K, K],
             coupling=dict(lin_coupled_modes=[1, 0, 0, 1, 0, 0], coupling_type=[0], coupl_trafo_matrices=[None] * 6),
             constrained_modes=[1, 1, 1, 1, 0, 1],
             constraints=[('non-negativity',)] * 4 + [None, ('non-negativity',)], weights=[0.5, 0.5], object=[X1, Xk])
    distr = [lambda a, b: rng.random((a, b))] * 4 + [lambda a, b: rng.standard_normal((a, b)), lambda a, b: rng.random((a, b))]
    io = dict(lambdas_init=[[1] * R, [1] * R], nvecs=0, distr=distr, normalize=1)
    return Z, io


def transformed_coupling_model(rng, ctype, noise=0.05):
    """Two CP tensors whose first modes are linearly coupled with transformation matrices (coupling types 1, 2, 3, 5:
    example_script5 / example_script13 families, scaled down): every-second-sample matrices on the row side
    (types 1, 3, 5), partially shared components on the column side (types 2, 5)."""
    n1, n4 = 25, 50
    sub = np.zeros((n1, n4))
    sub[np.arange(n1), 2 * np.arange(n1)] = 1.0                 # take every second entry (example_script5:41-45)
    if ctype in (1, 3):
        R1 = R4 = 3
    else:
        R1, R4 = 4, 3
    D = rng.random((n4, R1))
    if ctype in (1, 3, 5):
        A1 = sub @ D
        A4 = D[:, :R4]
    else:                                                       # type 2: same rows, shared columns
        n1 = n4 = 24
        D = rng.random((n1, R1))
        A1, A4 = D, D[:, :R4]
    A = [A1, rng.standard_normal((18, R1)), rng.random((20, R1))]
    B = [A4, rng.random((16, R4)), rng.standard_normal((14, R4))]
    X1, X2 = full_ktensor(A), full_ktensor(B)
    for X in (X1, X2):
        N = rng.standard_normal(X.shape)
        X += noise * np.linalg.norm(X) / np.linalg.norm(N) * N
    X1 /= np.linalg.norm(X1)
    X2 /= np.linalg.norm(X2)
    H = [None] * 6
    H2 = [None] * 6
    if ctype == 1:                                              # H*C = Delta
        H[0], H[3] = np.eye(n1), sub
    elif ctype == 2:                                            # C*H = Delta
        H[0], H[3] = np.vstack([np.eye(3), np.zeros((1, 3))]), np.eye(3)
    elif ctype == 3:                                            # C = H*Delta
        H[0], H[3] = sub, np.eye(n4)
    else:                                                       # H*C = Delta*H2 (example_script13:43-51)
        H[0], H[3] = np.eye(n1), sub
        H2[0], H2[3] = np.eye(4), np.vstack([np.eye(3), np.zeros((1, 3))])
    sz = [A[0].shape[0], 18, 20, B[0].shape[0], 16, 14]
    Z = dict(loss_function=['Frobenius'] * 2, model=['CP', 'CP'], modes=[[1, 2, 3], [4, 5, 6]], size=sz,
             coupling=dict(lin_coupled_modes=[1, 0, 0, 1, 0, 0], coupling_type=[ctype], coupl_trafo_matrices=H,
                           coupl_trafo_matrices2=H2),
             constrained_modes=[1, 0, 1, 1, 1, 0],
             constraints=[('non-negativity',), None, ('non-negativity',), ('non-negativity',), ('non-negativity',), None],
             weights=[0.5, 0.5], object=[X1, X2])
    distr = [lambda a, b: rng.random((a, b))] * 6
    io = dict(lambdas_init=[[1] * R1, [1] * R4], nvecs=0, distr=distr, normalize=1)
    return Z, io


def par2_C_coupled_model(rng, ctype, noise=0.0, K=16, I2=12, Jk=14, average=False):
    """example_script14_CP_PAR2_couplC_doublesamplingrate.m:20-44 scaled down: a CP tensor whose first mode is coupled
    to the C mode (third mode) of a PARAFAC2 block.  ctype 1: H{1}*C1 = Delta = H{6}*C6 with H{6} taking every second
    of the K rows (script 14); ctype 0: C1 = C6 exactly (K rows each)."""
    R = 3
    C6 = rng.random((K, R)) + 0.1
    if ctype == 1:
        n1 = K // 2
        sub = np.zeros((n1, K))
        sub[np.arange(n1), 2 * np.arange(n1)] = 1.0
        if average:                                 # means of neighbouring rows: H'H is not diagonal (dense (K*R)-system)
            sub[np.arange(n1), 2 * np.arange(n1)] = 0.5
            sub[np.arange(n1), 2 * np.arange(n1) + 1] = 0.5
        C1 = sub @ C6
        H = [np.eye(n1), None, None, None, None, sub]
    else:
        n1 = K
        C1 = C6.copy()
        H = [None] * 6
    X1 = full_ktensor([C1, rng.random((15, R)), rng.random((13, R))])
    A4 = rng.random((I2, R))
    DB = rng.standard_normal((R, R))
    Xk = []
    for k in range(K):
        Q, _ = np.linalg.qr(rng.standard_normal((Jk, R)))
        Xk.append(A4 @ np.diag(C6[k]) @ (Q @ DB).T)
    if noise > 0:
        N = rng.standard_normal(X1.shape)
        X1 = X1 + noise * np.linalg.norm(X1) / np.linalg.norm(N) * N
        Xk = [x + noise * np.linalg.norm(x) / np.sqrt(x.size) * rng.standard_normal(x.shape) for x in Xk]
    X1 /= np.linalg.norm(X1)
    nrm = np.sqrt(sum(np.linalg.norm(x) ** 2 for x in Xk))
    Xk = [x / nrm for x in Xk]
    Z = dict(loss_function=['Frobenius'] * 2, model=['CP', 'PAR2'], modes=[[1, 2, 3], [4, 5, 6]],
             size=[n1, 15, 13, I2, [Jk] * K, K],
             coupling=dict(lin_coupled_modes=[1, 0, 0, 0, 0, 1], coupling_type=[ctype], coupl_trafo_matrices=H),
             constrained_modes=[1, 1, 1, 1, 0, 1],
             constraints=[('non-negativity',)] * 4 + [None, ('non-negativity',)], weights=[0.5, 0.5], object=[X1, Xk])
    distr = [lambda a, b: rng.random((a, b)) + 0.1] + [lambda a, b: rng.random((a, b))] * 3 + \
            [lambda a, b: rng.standard_normal((a, b)), lambda a, b: rng.random((a, b)) + 0.1]
    io = dict(lambdas_init=[[1] * R, [1] * R], nvecs=0, distr=distr, normalize=1)
    return Z, io


def par2_C_transformed_model(rng, ctype, noise=0.05, K=14, I2=10, Jk=12):
    """CP first mode coupled to a PARAFAC2 C mode through a transformation of type 2 (C*H = Delta, shared columns),
    3 (C = H*Delta, row sampling), 4 (C = Delta*H, shared columns) or 5 (H*C = Delta*H2, both) --
    cmtf_fun_AOADMM.m:300-385, :777-1075."""
    R6 = 3
    if ctype == 3:
        R1 = 3
        D = rng.random((K, R6)) + 0.1
        n1 = K // 2
        sub = np.zeros((n1, K))
        sub[np.arange(n1), 2 * np.arange(n1)] = 1.0
        C6, C1 = D, sub @ D
        H = [sub, None, None, None, None, np.eye(K)]
    elif ctype == 5:                                               # H*C = Delta*H2 (example_script13 relations)
        R1 = 4
        D = rng.random((K, R1)) + 0.1
        n1 = K // 2
        sub = np.zeros((n1, K))
        sub[np.arange(n1), 2 * np.arange(n1)] = 1.0
        C6, C1 = D[:, :R6], sub @ D                                # Delta = sub*D (n1 x 4): C1 = Delta, sub*C6 = Delta(:,1:3)
        H = [np.eye(n1), None, None, None, None, sub]
        H2 = [np.eye(R1), None, None, None, None, np.vstack([np.eye(R6), np.zeros((1, R6))])]
    else:
        R1 = 4
        D = rng.random((K, R1)) + 0.1
        n1 = K
        C1, C6 = D, D[:, :R6]
        sel = np.vstack([np.eye(R6), np.zeros((1, R6))])          # 4 x 3
        if ctype == 2:                                             # C*H = Delta (K x 3)
            H = [sel, None, None, None, None, np.eye(R6)]
        else:                                                      # C = Delta*H, Delta is K x 4
            H = [np.eye(R1), None, None, None, None, sel]
    X1 = full_ktensor([C1, rng.random((13, R1)), rng.random((11, R1))])
    A4 = rng.random((I2, R6))
    DB = rng.standard_normal((R6, R6))
    Xk = []
    for k in range(K):
        Q, _ = np.linalg.qr(rng.standard_normal((Jk, R6)))
        Xk.append(A4 @ np.diag(C6[k]) @ (Q @ DB).T)
    if noise > 0:
        N = rng.standard_normal(X1.shape)
        X1 = X1 + noise * np.linalg.norm(X1) / np.linalg.norm(N) * N
        Xk = [x + noise * np.linalg.norm(x) / np.sqrt(x.size) * rng.standard_normal(x.shape) for x in Xk]
    X1 /= np.linalg.norm(X1)
    nrm = np.sqrt(sum(np.linalg.norm(x) ** 2 for x in Xk))
    Xk = [x / nrm for x in Xk]
    Z = dict(loss_function=['Frobenius'] * 2, model=['CP', 'PAR2'], modes=[[1, 2, 3], [4, 5, 6]],
             size=[n1, 13, 11, I2, [Jk] * K, K],
             coupling=dict(lin_coupled_modes=[1, 0, 0, 0, 0, 1], coupling_type=[ctype], coupl_trafo_matrices=H,
                           **({'coupl_trafo_matrices2': H2} if ctype == 5 else {})),
             constrained_modes=[1, 1, 0, 1, 0, 1],
             constraints=[('non-negativity',), ('non-negativity',), None, ('non-negativity',), None, ('non-negativity',)],
             weights=[0.5, 0.5], object=[X1, Xk])
    distr = [lambda a, b: rng.random((a, b)) + 0.1] + [lambda a, b: rng.random((a, b))] * 3 + \
            [lambda a, b: rng.standard_normal((a, b)), lambda a, b: rng.random((a, b)) + 0.1]
    io = dict(lambdas_init=[[1] * R1, [1] * R6], nvecs=0, distr=distr, normalize=1)
    return Z, io
