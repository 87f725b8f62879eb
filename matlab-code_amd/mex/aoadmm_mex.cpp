/*
 * aoadmm_mex.cpp -- MEX gateway: MATLAB <-> libaoadmm_hip.so (C ABI in include/aoadmm_hip.h).
 *
 * Marshalling only.  It replaces the call
 *     [Fac,out] = cmtf_fun_AOADMM(Z,Znorm_const,G,fh,gh,lscalar,uscalar,options)
 * at functions/cmtf_AOADMM.m:193 of the reference: the MATLAB wrapper
 * cmtf_fun_AOADMM_hip.m (same directory) passes the structs Z, G and options; this
 * file walks them with the mx* API, feeds the engine through the C ABI and builds
 * `Fac` (same fields as G) and `out` (cmtf_fun_AOADMM.m:480-494).
 *
 * Build (on a machine that has MATLAB; neither mex.h nor libmx exist in the
 * development container, where this file is only syntax-checked against
 * mex_stub/mex.h by `make -C matlab-code_amd/mex check`):
 *     mex -R2018a -largeArrayDims aoadmm_mex.cpp -I../../include -L.. -laoadmm_hip
 *
 * Errors: every non-zero status becomes mexErrMsgIdAndTxt with an id in the
 * `cmtf:` namespace; AOADMM_ERR_UNSUPPORTED maps to
 * `cmtf:hip:unsupported`, which the wrapper catches to fall back to the original
 * MATLAB implementation.
 */
#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#include "mex.h"
#include "aoadmm_hip.h"

namespace {

aoadmm_ctx* g_ctx = nullptr;
std::vector<int> g_devices;      // the device list g_ctx was created for
bool g_at_exit = false;

void at_exit() {
  if (g_ctx) {
    aoadmm_destroy(g_ctx);
    g_ctx = nullptr;
  }
}

void check(int status) {
  if (status == AOADMM_OK) return;
  const char* msg = aoadmm_last_error();
  switch (status) {
    case AOADMM_ERR_NOT_PD: mexErrMsgIdAndTxt("cmtf:hip:notPositiveDefinite", "%s", msg); break;
    case AOADMM_ERR_UNSUPPORTED: mexErrMsgIdAndTxt("cmtf:hip:unsupported", "%s", msg); break;
    case AOADMM_ERR_HIP: mexErrMsgIdAndTxt("cmtf:hip:device", "%s", msg); break;
    case AOADMM_ERR_RCCL: mexErrMsgIdAndTxt("cmtf:hip:rccl", "%s", msg); break;
    case AOADMM_ERR_NOMEM: mexErrMsgIdAndTxt("cmtf:hip:outOfMemory", "%s", msg); break;
    default: mexErrMsgIdAndTxt("cmtf:hip:invalid", "%s", msg); break;
  }
}

const mxArray* field(const mxArray* s, const char* name, bool required = true) {
  const mxArray* f = mxIsStruct(s) ? mxGetField(s, 0, name) : nullptr;
  if (!f && required) mexErrMsgIdAndTxt("cmtf:hip:missingField", "Reference to non-existent field '%s'.", name);
  return f;
}

double scalar(const mxArray* s, const char* name) { return mxGetScalar(field(s, name)); }

std::string str(const mxArray* a) {
  char* c = mxArrayToString(a);
  std::string r = c ? c : "";
  mxFree(c);
  return r;
}

// constraint names of "List of constraints and regularizations.txt" -> AOADMM_C_* ids
int constraint_id(const std::string& n) {
  static const struct { const char* name; int id; } tab[] = {
      {"non-negativity", AOADMM_C_NONNEG}, {"box", AOADMM_C_BOX}, {"simplex column-wise", AOADMM_C_SIMPLEX_COL},
      {"simplex row-wise", AOADMM_C_SIMPLEX_ROW}, {"non-decreasing", AOADMM_C_NONDECREASING},
      {"non-increasing", AOADMM_C_NONINCREASING}, {"unimodality", AOADMM_C_UNIMODAL}, {"l1-ball", AOADMM_C_L1_BALL},
      {"l2-ball", AOADMM_C_L2_BALL}, {"non-negative l2-ball", AOADMM_C_NONNEG_L2_BALL},
      {"non-negative l2-sphere", AOADMM_C_NONNEG_L2_SPHERE}, {"orthonormal", AOADMM_C_ORTHONORMAL},
      {"l1 regularization", AOADMM_C_L1_REG}, {"l0 regularization", AOADMM_C_L0_REG},
      {"l2 regularization", AOADMM_C_L2_REG}, {"ridge", AOADMM_C_RIDGE},
      {"quadratic regularization", AOADMM_C_QUADRATIC}, {"GL smoothness", AOADMM_C_GL_SMOOTH},
      {"TV regularization", AOADMM_C_TV}, {"tPARAFAC2", AOADMM_C_TPARAFAC2}};
  for (const auto& e : tab)
    if (n == e.name) return e.id;
  if (n == "custom")
    mexErrMsgIdAndTxt("cmtf:hip:unsupported", "'custom' prox handles cannot cross to the device (constraints_to_prox.m:86-90)");
  mexErrMsgIdAndTxt("cmtf:hip:invalid", "unknown constraint '%s'", n.c_str());
  return 0;
}

// dense numeric data of a Tensor Toolbox `tensor` (field .data) or a plain array
const mxArray* dense_data(const mxArray* obj) {
  if (mxIsDouble(obj)) return obj;
  const mxArray* d = mxIsClass(obj, "tensor") ? mxGetProperty(obj, 0, "data") : nullptr;
  if (!d || !mxIsDouble(d))
    mexErrMsgIdAndTxt("cmtf:hip:unsupported", "Z.object must be a dense double array or a dense tensor (sptensor stays on the MATLAB path)");
  return d;
}

void put_state(int field_id, int index, int slab, const mxArray* a) {
  if (!a || mxIsEmpty(a)) return;
  check(aoadmm_state_set(g_ctx, field_id, index, slab, mxGetDoubles(a), (int64_t)mxGetM(a), (int64_t)mxGetN(a)));
}

void put_state_maybe_cell(int field_id, int index, const mxArray* a) {
  if (!a || mxIsEmpty(a)) return;
  if (mxIsCell(a)) {
    // all cells in one transfer: J_k x R blocks back to back (AOADMM_ALL_SLABS)
    std::vector<double> packed;
    int64_t rows = 0, cols = 0;
    for (mwSize k = 0; k < mxGetNumberOfElements(a); ++k) {
      const mxArray* ak = mxGetCell(a, k);
      if (!ak || mxIsEmpty(ak)) mexErrMsgIdAndTxt("cmtf:hip:state", "empty cell %d in a cell-valued state field", (int)k + 1);
      const double* d = mxGetDoubles(ak);
      packed.insert(packed.end(), d, d + mxGetNumberOfElements(ak));
      rows += (int64_t)mxGetM(ak);
      cols = (int64_t)mxGetN(ak);
    }
    check(aoadmm_state_set(g_ctx, field_id, index, AOADMM_ALL_SLABS, packed.data(), rows, cols));
  } else {
    put_state(field_id, index, 0, a);
  }
}

struct ProgressState { bool has_missing = false; };
void print_progress(void* user, int iter, const double f[4], double f_rel_missing) {
  const ProgressState* ps = static_cast<const ProgressState*>(user);
  if (ps->has_missing)
    mexPrintf("%6d %12f %12f %12f %17f %12f %12f\n", iter, f[0] + f[1] + f[2] + f[3], f[0], f[1], f[2], f[3], f_rel_missing);
  else
    mexPrintf("%6d %12f %12f %12f %17f %12f\n", iter, f[0] + f[1] + f[2] + f[3], f[0], f[1], f[2], f[3]);
  mexEvalString("drawnow;");                          // flush the command window while the solve is running
}

mxArray* get_like(int field_id, int index, const mxArray* ref) {
  if (!ref || mxIsEmpty(ref)) return mxCreateDoubleMatrix(0, 0, mxREAL);
  if (mxIsCell(ref)) {
    mxArray* c = mxCreateCellMatrix(mxGetM(ref), mxGetN(ref));
    int64_t rows = 0, cols = 0;
    for (mwSize k = 0; k < mxGetNumberOfElements(ref); ++k) {
      rows += (int64_t)mxGetM(mxGetCell(ref, k));
      cols = (int64_t)mxGetN(mxGetCell(ref, k));
    }
    std::vector<double> packed((size_t)(rows * cols));
    check(aoadmm_state_get(g_ctx, field_id, index, AOADMM_ALL_SLABS, packed.data(), rows, cols));
    size_t o0 = 0;
    for (mwSize k = 0; k < mxGetNumberOfElements(ref); ++k) {
      const mxArray* rk = mxGetCell(ref, k);
      mxArray* o = mxCreateDoubleMatrix(mxGetM(rk), mxGetN(rk), mxREAL);
      std::copy(packed.begin() + o0, packed.begin() + o0 + mxGetNumberOfElements(rk), mxGetDoubles(o));
      o0 += mxGetNumberOfElements(rk);
      mxSetCell(c, k, o);
    }
    return c;
  }
  mxArray* o = mxCreateDoubleMatrix(mxGetM(ref), mxGetN(ref), mxREAL);
  check(aoadmm_state_get(g_ctx, field_id, index, 0, mxGetDoubles(o), (int64_t)mxGetM(ref), (int64_t)mxGetN(ref)));
  return o;
}

}  // namespace

/* [Fac, out] = aoadmm_mex(Z, G, options)      (options.hip.device / .precision are optional) */
void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
  // Y = aoadmm_mex('unfold_gram', X, n): Gram matrix of the mode-n unfolding (cmtf_nvecs.m:40-56), n 1-based
  if (nrhs == 3 && mxIsChar(prhs[0])) {
    if (str(prhs[0]) != "unfold_gram") mexErrMsgIdAndTxt("cmtf:hip:usage", "unknown operation '%s'", str(prhs[0]).c_str());
    if (!g_ctx) {
      check(aoadmm_create(&g_ctx, 0));
      g_devices.assign(1, 0);
      if (!g_at_exit) { mexAtExit(at_exit); g_at_exit = true; }
    }
    const mxArray* X = prhs[1];
    const int nd = (int)mxGetNumberOfDimensions(X);
    const mwSize* d = mxGetDimensions(X);
    int64_t dims[8];
    for (int i = 0; i < nd && i < 8; ++i) dims[i] = (int64_t)d[i];
    const int n = (int)mxGetScalar(prhs[2]) - 1;
    if (nd < 2 || nd > 3 || n < 0 || n >= nd) mexErrMsgIdAndTxt("cmtf:hip:unsupported", "unfold_gram handles matrices and 3-way tensors");
    plhs[0] = mxCreateDoubleMatrix((mwSize)dims[n], (mwSize)dims[n], mxREAL);
    check(aoadmm_op_unfold_gram(g_ctx, mxGetDoubles(X), nd, dims, n, AOADMM_PREC_F64, mxGetDoubles(plhs[0])));
    return;
  }
  if (nrhs != 3 || nlhs > 2) mexErrMsgIdAndTxt("cmtf:hip:usage", "usage: [Fac,out] = aoadmm_mex(Z, G, options)");
  const mxArray *Z = prhs[0], *G = prhs[1], *opt = prhs[2];
  // options.hip.device = d (one GPU) or options.hip.devices = [d0 d1 ...] (several GPUs from this one MATLAB process:
  // aoadmm_create_multi, one engine + host thread per device inside the library, RCCL between them)
  std::vector<int> devices(1, 0);
  int precision = AOADMM_PREC_F64;
  if (const mxArray* hip = field(opt, "hip", false)) {
    if (const mxArray* d = field(hip, "device", false)) devices.assign(1, (int)mxGetScalar(d));
    if (const mxArray* d = field(hip, "devices", false)) {
      devices.clear();
      for (mwSize i = 0; i < mxGetNumberOfElements(d); ++i) devices.push_back((int)mxGetDoubles(d)[i]);
      if (devices.empty()) devices.assign(1, 0);
    }
    if (const mxArray* p = field(hip, "precision", false)) precision = str(p) == "f32" ? AOADMM_PREC_F32 : AOADMM_PREC_F64;
  }
  if (g_ctx && devices != g_devices) {               // another device list: start over
    aoadmm_destroy(g_ctx);
    g_ctx = nullptr;
  }
  if (!g_ctx) {
    if (devices.size() == 1) check(aoadmm_create(&g_ctx, devices[0]));
    else check(aoadmm_create_multi(&g_ctx, (int)devices.size(), devices.data()));
    g_devices = devices;
    if (!g_at_exit) { mexAtExit(at_exit); g_at_exit = true; }
  }

  // ---- model: Z.size, Z.modes, Z.model, Z.weights, Z.coupling, Z.constraints (example_script1:74-89)
  const mxArray* sz = field(Z, "size");
  const mxArray* modes = field(Z, "modes");
  const mxArray* model = field(Z, "model");
  const mxArray* weights = field(Z, "weights");
  const mxArray* coupling = field(Z, "coupling");
  const mxArray* lin = field(coupling, "lin_coupled_modes");
  const mxArray* ctype = field(coupling, "coupling_type");
  const mxArray* trafo = field(coupling, "coupl_trafo_matrices");
  const mxArray* trafo2 = field(coupling, "coupl_trafo_matrices2", false);
  const mxArray* cmodes = field(Z, "constrained_modes");
  const mxArray* constraints = field(Z, "constraints");
  const mxArray* object = field(Z, "object");
  const mxArray* fac = field(G, "fac");
  const int n_modes = (int)mxGetNumberOfElements(sz);
  const int P = (int)mxGetNumberOfElements(object);
  const double* linv = mxGetDoubles(lin);
  int n_couplings = 0;
  for (int m = 0; m < n_modes; ++m) n_couplings = linv[m] > n_couplings ? (int)linv[m] : n_couplings;
  for (int p = 0; p < P; ++p)
    if (str(mxGetCell(field(Z, "loss_function"), p)) != "Frobenius")
      mexErrMsgIdAndTxt("cmtf:hip:unsupported", "non-Frobenius losses need the L-BFGS-B path of the MATLAB code");
  const mxArray* miss = field(Z, "miss", false);        // cmtf_fun_AOADMM_hip.m passes dense uint8 masks (1 = observed)
  bool has_missing = false;

  check(aoadmm_model_begin(g_ctx, n_modes, P, n_couplings));
  for (int m = 0; m < n_modes; ++m) {
    const mxArray* s = mxGetCell(sz, m);
    const mxArray* f = mxGetCell(fac, m);
    const int R = (int)mxGetN(mxIsCell(f) ? mxGetCell(f, 0) : f);
    if (mxGetNumberOfElements(s) > 1) {
      std::vector<int64_t> rows(mxGetNumberOfElements(s));
      for (size_t k = 0; k < rows.size(); ++k) rows[k] = (int64_t)mxGetDoubles(s)[k];
      check(aoadmm_model_set_mode_slabs(g_ctx, m, (int)rows.size(), rows.data(), R));
    } else {
      check(aoadmm_model_set_mode(g_ctx, m, (int64_t)mxGetScalar(s), R));
    }
  }
  for (int p = 0; p < P; ++p) {
    const mxArray* mp = mxGetCell(modes, p);
    std::vector<int> md(mxGetNumberOfElements(mp));
    for (size_t i = 0; i < md.size(); ++i) md[i] = (int)mxGetDoubles(mp)[i] - 1;
    const double w = mxGetDoubles(weights)[p];
    if (str(mxGetCell(model, p)) == "CP") check(aoadmm_model_add_cp(g_ctx, p, (int)md.size(), md.data(), w));
    else check(aoadmm_model_add_par2(g_ctx, p, md.data(), w));
  }
  for (int m = 0; m < n_modes; ++m) {
    if (mxGetDoubles(cmodes)[m] != 0) {
      const mxArray* c = mxGetCell(constraints, m);
      if (!c || mxIsEmpty(c)) mexErrMsgIdAndTxt("cmtf:hip:invalid", "No constraint provided for mode %d.", m + 1);
      const int id = constraint_id(str(mxGetCell(c, 0)));
      std::vector<double> par;
      const double* Lmat = nullptr;
      for (mwSize q = 1; q < mxGetNumberOfElements(c); ++q) {
        const mxArray* v = mxGetCell(c, q);
        if (id == AOADMM_C_QUADRATIC && q == 2) Lmat = mxGetDoubles(v);
        else par.push_back(mxGetScalar(v));
      }
      check(aoadmm_model_set_constraint(g_ctx, m, id, par.data(), (int)par.size(), Lmat));
    }
    const mxArray* H = mxGetCell(trafo, m);
    const mxArray* H2 = trafo2 ? mxGetCell(trafo2, m) : nullptr;
    const bool coupled = linv[m] > 0;
    check(aoadmm_model_set_coupling(g_ctx, m, (int)linv[m] - 1,
                                    coupled && H && !mxIsEmpty(H) ? mxGetDoubles(H) : nullptr, H ? (int64_t)mxGetM(H) : 0,
                                    H ? (int64_t)mxGetN(H) : 0,
                                    coupled && H2 && !mxIsEmpty(H2) ? mxGetDoubles(H2) : nullptr,
                                    H2 ? (int64_t)mxGetM(H2) : 0, H2 ? (int64_t)mxGetN(H2) : 0));
  }
  for (int c = 0; c < n_couplings; ++c) check(aoadmm_model_set_coupling_type(g_ctx, c, (int)mxGetDoubles(ctype)[c]));
  if (const mxArray* ridge = field(Z, "ridge", false)) check(aoadmm_model_set_ridge(g_ctx, mxGetDoubles(ridge)));
  check(aoadmm_model_end(g_ctx));

  // ---- data: Z.object{p}
  for (int p = 0; p < P; ++p) {
    const mxArray* obj = mxGetCell(object, p);
    if (mxIsCell(obj)) {
      std::vector<double> packed;                         // the slabs back to back, one transfer
      for (mwSize k = 0; k < mxGetNumberOfElements(obj); ++k) {
        const mxArray* xk = mxGetCell(obj, k);
        packed.insert(packed.end(), mxGetDoubles(xk), mxGetDoubles(xk) + mxGetNumberOfElements(xk));
      }
      check(aoadmm_par2_slab_upload(g_ctx, p, AOADMM_ALL_SLABS, packed.data()));
    } else {
      check(aoadmm_tensor_upload(g_ctx, p, mxGetDoubles(dense_data(obj)), precision));
    }
    // Z.miss{p} (cmtf_AOADMM.m:68-121): same shape as the data, uint8
    const mxArray* mk = (miss && !mxIsEmpty(miss) && (mwSize)p < mxGetNumberOfElements(miss)) ? mxGetCell(miss, p) : nullptr;
    if (mk && !mxIsEmpty(mk)) {
      has_missing = true;
      if (mxIsCell(obj)) {
        if (!mxIsCell(mk) || mxGetNumberOfElements(mk) != mxGetNumberOfElements(obj))
          mexErrMsgIdAndTxt("cmtf:missingData:PAR2maskNotCell", "Z.miss{%d} must be a cell array of length %d for PAR2.", p + 1,
                            (int)mxGetNumberOfElements(obj));
        for (mwSize k = 0; k < mxGetNumberOfElements(obj); ++k) {
          const mxArray* mkk = mxGetCell(mk, k);
          if (!mxIsUint8(mkk) || mxGetNumberOfElements(mkk) != mxGetNumberOfElements(mxGetCell(obj, k)))
            mexErrMsgIdAndTxt("cmtf:missingData:PAR2maskSliceSizeMismatch", "Z.miss{%d}{%d} size does not match Z.object{%d}{%d}.",
                              p + 1, (int)k + 1, p + 1, (int)k + 1);
          check(aoadmm_par2_slab_mask_upload(g_ctx, p, (int)k, static_cast<const uint8_t*>(mxGetData(mkk))));
        }
      } else {
        if (!mxIsUint8(mk) || mxGetNumberOfElements(mk) != mxGetNumberOfElements(dense_data(obj)))
          mexErrMsgIdAndTxt("cmtf:missingData:maskSizeMismatch", "Z.miss{%d} size does not match Z.object{%d}.", p + 1, p + 1);
        check(aoadmm_tensor_mask_upload(g_ctx, p, static_cast<const uint8_t*>(mxGetData(mk))));
      }
    }
  }

  // ---- state: the struct G (init_coupled_AOADMM_CMTF.m:41-45)
  for (int m = 0; m < n_modes; ++m) {
    put_state_maybe_cell(AOADMM_F_FAC, m, mxGetCell(fac, m));
    if (const mxArray* f = field(G, "constraint_fac", false)) put_state_maybe_cell(AOADMM_F_CONSTRAINT_FAC, m, mxGetCell(f, m));
    if (const mxArray* f = field(G, "constraint_dual_fac", false)) put_state_maybe_cell(AOADMM_F_CONSTRAINT_DUAL, m, mxGetCell(f, m));
    if (const mxArray* f = field(G, "coupling_dual_fac", false)) put_state_maybe_cell(AOADMM_F_COUPLING_DUAL, m, mxGetCell(f, m));
  }
  if (const mxArray* f = field(G, "coupling_fac", false))
    for (int c = 0; c < n_couplings; ++c) put_state(AOADMM_F_COUPLING_FAC, c, 0, mxGetCell(f, c));
  for (int p = 0; p < P; ++p) {
    if (const mxArray* f = field(G, "DeltaB", false))
      if ((mwSize)p < mxGetNumberOfElements(f)) put_state(AOADMM_F_DELTAB, p, 0, mxGetCell(f, p));
    if (const mxArray* f = field(G, "P", false))
      if ((mwSize)p < mxGetNumberOfElements(f)) put_state_maybe_cell(AOADMM_F_P, p, mxGetCell(f, p));
    if (const mxArray* f = field(G, "mu_DeltaB", false))
      if ((mwSize)p < mxGetNumberOfElements(f)) put_state_maybe_cell(AOADMM_F_MU_DELTAB, p, mxGetCell(f, p));
  }

  // ---- options (example_script1_CP_PAR2_nonneg.m:110-123; missing fields error like MATLAB)
  aoadmm_options o;
  std::memset(&o, 0, sizeof o);
  o.MaxOuterIters = (int)scalar(opt, "MaxOuterIters");
  o.MaxInnerIters = (int)scalar(opt, "MaxInnerIters");
  o.AbsFuncTol = scalar(opt, "AbsFuncTol");
  o.OuterRelTol = scalar(opt, "OuterRelTol");
  o.innerRelPrTol_coupl = scalar(opt, "innerRelPrTol_coupl");
  o.innerRelPrTol_constr = scalar(opt, "innerRelPrTol_constr");
  o.innerRelDualTol_coupl = scalar(opt, "innerRelDualTol_coupl");
  o.innerRelDualTol_constr = scalar(opt, "innerRelDualTol_constr");
  o.bsum = scalar(opt, "bsum") != 0;
  if (o.bsum) o.bsum_weight = scalar(opt, "bsum_weight");
  if (const mxArray* f = field(opt, "iter_start_PAR2Bkconstraint", false)) o.iter_start_PAR2Bkconstraint = (int)mxGetScalar(f);
  if (const mxArray* f = field(opt, "increase_factor_rhoBk", false)) {
    o.has_increase_factor_rhoBk = 1;
    o.increase_factor_rhoBk = mxGetScalar(f);
  }
  o.use_dimtree = 1;
  if (const mxArray* hip = field(opt, "hip", false)) {
    if (const mxArray* f = field(hip, "no_permuted_copy", false)) o.no_permuted_copy = (int)mxGetScalar(f);
    if (const mxArray* f = field(hip, "par2_slab_sharding", false)) o.par2_slab_sharding = (int)mxGetScalar(f);
  }

  // ---- solve
  const int n = o.MaxOuterIters + 1;
  std::vector<double> fv(n), fc(n), fz(n), fp(n), tt(n), frm(n), inner((size_t)n_modes * (o.MaxOuterIters > 0 ? o.MaxOuterIters : 1));
  aoadmm_result res;
  std::memset(&res, 0, sizeof res);
  res.func_val_conv = fv.data(); res.func_coupl_conv = fc.data(); res.func_constr_conv = fz.data();
  res.func_PAR2_coupl = fp.data(); res.time_at_it = tt.data(); res.innerIters = inner.data();
  res.func_rel_missing = frm.data();
  // options.Display = 'iter': one table row every DisplayIters iterations, live (cmtf_fun_AOADMM.m:53-59, :462-468)
  ProgressState ps;
  ps.has_missing = has_missing;
  const mxArray* disp = field(opt, "Display", false);
  const bool live = disp && mxIsChar(disp) && str(disp) == "iter";
  if (live) {
    const mxArray* di = field(opt, "DisplayIters", false);
    check(aoadmm_set_progress(g_ctx, &print_progress, &ps, di ? (int)mxGetScalar(di) : 10));
  }
  const int rc = aoadmm_solve(g_ctx, &o, &res);
  if (live) (void)aoadmm_set_progress(g_ctx, nullptr, nullptr, 0);
  check(rc);

  // ---- Fac: same fields as G (cmtf_AOADMM.m:193,197-206)
  plhs[0] = mxDuplicateArray(G);
  mxArray* Fac = plhs[0];
  for (int m = 0; m < n_modes; ++m) {
    mxSetCell(mxGetField(Fac, 0, "fac"), m, get_like(AOADMM_F_FAC, m, mxGetCell(fac, m)));
    const char* names[3] = {"constraint_fac", "constraint_dual_fac", "coupling_dual_fac"};
    const int ids[3] = {AOADMM_F_CONSTRAINT_FAC, AOADMM_F_CONSTRAINT_DUAL, AOADMM_F_COUPLING_DUAL};
    for (int q = 0; q < 3; ++q)
      if (mxArray* f = mxGetField(Fac, 0, names[q])) {
        const mxArray* ref = mxGetCell(f, m);
        if (ref && !mxIsEmpty(ref)) mxSetCell(f, m, get_like(ids[q], m, ref));
      }
  }
  if (mxArray* f = mxGetField(Fac, 0, "coupling_fac"))
    for (int c = 0; c < n_couplings; ++c) mxSetCell(f, c, get_like(AOADMM_F_COUPLING_FAC, c, mxGetCell(f, c)));
  for (int p = 0; p < P; ++p) {
    const char* names[3] = {"DeltaB", "P", "mu_DeltaB"};
    const int ids[3] = {AOADMM_F_DELTAB, AOADMM_F_P, AOADMM_F_MU_DELTAB};
    for (int q = 0; q < 3; ++q)
      if (mxArray* f = mxGetField(Fac, 0, names[q]))
        if ((mwSize)p < mxGetNumberOfElements(f)) {
          const mxArray* ref = mxGetCell(f, p);
          if (ref && !mxIsEmpty(ref)) mxSetCell(f, p, get_like(ids[q], p, ref));
        }
  }

  // ---- out (cmtf_fun_AOADMM.m:480-494)
  if (nlhs > 1) {
    const char* fn[] = {"f_tensors", "f_couplings", "f_constraints", "f_PAR2_couplings", "f_rel_missing", "exit_flag",
                        "OuterIterations", "func_val_conv", "func_coupl_conv", "func_constr_conv", "func_PAR2_coupl",
                        "time_at_it", "innerIters", "func_rel_missing"};
    mxArray* out = mxCreateStructMatrix(1, 1, has_missing ? 14 : 13, fn);   /* func_rel_missing only with Z.miss (:490-492) */
    const int it = res.OuterIterations;
    auto vec = [&](const std::vector<double>& v, int len) {
      mxArray* a = mxCreateDoubleMatrix(1, len, mxREAL);
      std::memcpy(mxGetDoubles(a), v.data(), sizeof(double) * len);
      return a;
    };
    mxSetField(out, 0, "f_tensors", mxCreateDoubleScalar(res.f_tensors));
    mxSetField(out, 0, "f_couplings", mxCreateDoubleScalar(res.f_couplings));
    mxSetField(out, 0, "f_constraints", mxCreateDoubleScalar(res.f_constraints));
    mxSetField(out, 0, "f_PAR2_couplings", mxCreateDoubleScalar(res.f_PAR2_couplings));
    mxSetField(out, 0, "f_rel_missing", mxCreateDoubleScalar(has_missing ? res.f_rel_missing : mxGetNaN()));
    if (has_missing) mxSetField(out, 0, "func_rel_missing", vec(frm, it + 1));
    if (res.exit_code == 0) {
      mxSetField(out, 0, "exit_flag", mxCreateString("maxIterations"));      /* make_exit_flag.m:4-5 */
    } else {
      const char* q[] = {"f_tensors", "f_couplings", "f_constraints", "f_PAR2_couplings"};
      mxArray* ef = mxCreateStructMatrix(1, 1, 4, q);
      for (int i = 0; i < 4; ++i) mxSetField(ef, 0, q[i], mxCreateString(res.exit_abs[i] ? "AbsFuncTol" : "RelFuncTol"));
      mxSetField(out, 0, "exit_flag", ef);
    }
    mxSetField(out, 0, "OuterIterations", mxCreateDoubleScalar(it));
    mxSetField(out, 0, "func_val_conv", vec(fv, it + 1));
    mxSetField(out, 0, "func_coupl_conv", vec(fc, it + 1));
    mxSetField(out, 0, "func_constr_conv", vec(fz, it + 1));
    mxSetField(out, 0, "func_PAR2_coupl", vec(fp, it + 1));
    mxSetField(out, 0, "time_at_it", vec(tt, it + 1));
    mxArray* ii = mxCreateDoubleMatrix(n_modes, it > 0 ? it : 1, mxREAL);
    std::memcpy(mxGetDoubles(ii), inner.data(), sizeof(double) * n_modes * (it > 0 ? it : 1));
    mxSetField(out, 0, "innerIters", ii);
    plhs[1] = out;
  }
}
