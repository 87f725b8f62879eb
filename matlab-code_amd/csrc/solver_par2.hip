// Engine: PARAFAC2 block updates (functions/cmtf_fun_AOADMM.m:157-250, :509-589) on the device.
#include <algorithm>
#include <cstring>

#include "solver.h"

namespace aoadmm {

static constexpr int kSlotsPerMode = 8;
static constexpr int kResidPerMode = 8;

double* Engine::resid_slots(int m) {
  return slots_.d() + n_modes_ * kSlotsPerMode + 2 * n_tensors_ + (int64_t)m * kResidPerMode;
}

void Engine::add_par2(int p, const int* modes3, double weight) {
  AO_REQUIRE(p >= 0 && p < n_tensors_, "tensor %d out of range", p);
  TensorInfo& t = tensors_[p];
  t.defined = true; t.par2 = true; t.nmodes = 3; t.weight = weight;
  for (int i = 0; i < 3; ++i) {
    check_mode(modes3[i]);
    ModeInfo& mi = modes_[modes3[i]];
    AO_REQUIRE(mi.defined, "mode %d undefined", modes3[i]);
    AO_REQUIRE(mi.tensor < 0, "mode %d already belongs to tensor %d", modes3[i], mi.tensor);
    AO_REQUIRE(mi.slabs == (i == 1), "PARAFAC2: only the second mode is slab-valued (mode %d)", modes3[i]);
    t.modes[i] = modes3[i];
    mi.tensor = p;
    mi.pos = i;
  }
  const ModeInfo& mA = modes_[modes3[0]];
  const ModeInfo& mB = modes_[modes3[1]];
  const ModeInfo& mC = modes_[modes3[2]];
  // check_data_input.m:22-26: size of mode C must equal the number of slabs
  AO_REQUIRE(mC.rows == mB.K, "size mismatch in PARAFAC2 model betwwen mode C and Bk");
  AO_REQUIRE(mA.R == mB.R && mA.R == mC.R, "modes of tensor %d disagree on the rank", p);
  Par2Block& b = t.p2;
  b.K = mB.K; b.I = (int)mA.rows; b.R = mA.R;
  b.off_h = mB.off_k;
  b.Jtot = mB.rows;
  b.Jmax = 0;
  for (int k = 0; k < b.K; ++k) {
    // cmtf_AOADMM.m:55-65
    AO_REQUIRE(mB.rows_k[k] >= b.R, "Number of components for PARAFAC2 is larger than size of slice %d of data tensor %d.", k + 1, p + 1);
    b.Jmax = std::max<int>(b.Jmax, (int)mB.rows_k[k]);
  }
  b.off_d.alloc((size_t)(b.K + 1) * sizeof(int64_t));
  AO_HIP(hipMemcpyAsync(b.off_d.p, b.off_h.data(), (size_t)(b.K + 1) * sizeof(int64_t), hipMemcpyHostToDevice, stream_));
  AO_HIP(hipStreamSynchronize(stream_));
  b.X.alloc((size_t)b.I * b.Jtot * sizeof(double));
  b.have_slab.assign(b.K, 0);
  b.have_P.assign(b.K, 0);
  b.have_mu.assign(b.K, 0);
}

void Engine::par2_slab_upload(int p, int k, const double* Xk) {
  AO_REQUIRE(model_done_, "call aoadmm_model_end first");
  AO_REQUIRE(p >= 0 && p < n_tensors_ && tensors_[p].par2, "tensor %d is not a PARAFAC2 block", p);
  AO_REQUIRE(Xk != nullptr, "null slab");
  AO_HIP(hipSetDevice(device_));
  Par2Block& b = tensors_[p].p2;
  AO_REQUIRE(k == AOADMM_ALL_SLABS || (k >= 0 && k < b.K), "slab %d out of range", k);
  const bool all = k == AOADMM_ALL_SLABS;                  // I x sum(J_k): the slabs back to back
  const int64_t o = all ? 0 : b.off_h[k];
  const int64_t Jk = all ? b.Jtot : b.off_h[k + 1] - b.off_h[k];
  AO_HIP(hipMemcpyAsync(b.X.d() + (int64_t)b.I * o, Xk, (size_t)b.I * Jk * sizeof(double), hipMemcpyHostToDevice, stream_));
  AO_HIP(hipStreamSynchronize(stream_));
  if (all) b.have_slab.assign(b.K, 1); else b.have_slab[k] = 1;
  tensors_[p].blk.has_data = std::all_of(b.have_slab.begin(), b.have_slab.end(), [](char c) { return c != 0; });
  tensors_[p].normsq_valid = false;
}

void Engine::par2_slab_mask_upload(int p, int k, const uint8_t* mask) {
  AO_REQUIRE(model_done_, "call aoadmm_model_end first");
  AO_REQUIRE(p >= 0 && p < n_tensors_ && tensors_[p].par2, "tensor %d is not a PARAFAC2 block", p);
  AO_REQUIRE(mask != nullptr, "null mask");
  AO_HIP(hipSetDevice(device_));
  Par2Block& b = tensors_[p].p2;
  AO_REQUIRE(k == AOADMM_ALL_SLABS || (k >= 0 && k < b.K), "slab %d out of range", k);
  if (!b.has_mask) {
    b.mask.alloc((size_t)b.I * b.Jtot);
    AO_HIP(hipMemsetAsync(b.mask.p, 1, (size_t)b.I * b.Jtot, stream_));     // slabs without a mask: fully observed
    b.has_mask = true;
  }
  const int64_t o = k == AOADMM_ALL_SLABS ? 0 : b.off_h[k];
  const int64_t Jk = k == AOADMM_ALL_SLABS ? b.Jtot : b.off_h[k + 1] - b.off_h[k];
  AO_HIP(hipMemcpyAsync(b.mask.as<uint8_t>() + (int64_t)b.I * o, mask, (size_t)b.I * Jk, hipMemcpyHostToDevice, stream_));
  AO_HIP(hipStreamSynchronize(stream_));
  tensors_[p].normsq_valid = false;
}

void Engine::par2_ensure_work(TensorInfo& t) {
  Par2Block& b = t.p2;
  const size_t RR = (size_t)b.R * b.R * sizeof(double);
  const size_t cat = (size_t)b.Jtot * b.R * sizeof(double);
  b.DeltaBold.ensure(RR); b.Pold.ensure(cat); b.W.ensure(cat); b.Ak.ensure(cat);
  b.T1.ensure((size_t)b.K * b.I * b.R * sizeof(double));
  b.GB.ensure((size_t)b.K * RR); b.Lk.ensure((size_t)b.K * RR); b.Lc.ensure((size_t)b.K * RR);
  b.rhok.ensure((size_t)b.K * 8); b.rhoc.ensure((size_t)b.K * 8); b.rhomax.ensure(64);
  b.part.ensure((size_t)b.K * RR); b.norms.ensure((size_t)b.K * 8 * 8);
  b.res.ensure((size_t)(b.K + 1) * 8); b.q.ensure((size_t)b.K * 4 * 8); b.regv.ensure((size_t)b.K * 8);
  b.Csys.ensure(RR); b.ac.ensure((size_t)b.K * b.R * 8);
  b.psum.ensure(RR + 128);                       // R*R+1 sums of DeltaB, then (at R*R+8) four residual means
  ModeInfo& mB = modes_[t.modes[1]];
  const size_t nB = (size_t)mB.rows * mB.R * sizeof(double);
  mB.Zold.ensure(nB); mB.V.ensure(nB);
  size_t pw = 16;
  if (mB.constrained) pw = prox_ws_bytes(mB.prox.type, b.Jmax, b.R);
  mB.proxws.ensure(pw);
}

// mode A: MTTKRP and Hadamard analogue from the slabs (:160-178), then the common system build
void Engine::par2_prepare_modeA(int m, int nrho, const aoadmm_options& opt) {
  ModeInfo& mi = modes_[m];
  TensorInfo& t = tensors_[mi.tensor];
  Par2Block& b = t.p2;
  const P2Dims d = b.dims();
  ModeInfo& mB = modes_[t.modes[1]];
  ModeInfo& mC = modes_[t.modes[2]];
  par2_xkb(b.X.d(), mB.fac.d(), d, b.T1.d(), stream_);
  par2_modeA_combine(b.T1.d(), mC.fac.d(), b.GB.d(), d, mi.tmp.d(), b.Csys.d(), stream_);
  if (b.slab_sharded) {                               // sums over this rank's slabs -> sums over all slabs
    allreduce(mi.tmp.d(), mi.rows * mi.R);
    allreduce(b.Csys.d(), (int64_t)b.R * b.R);
  }
  {
    Coef c[1] = {coef(t.weight)};                     // A{m} = w*A{m}  (:169); last_mttkrp = A/w
    const double* x[1] = {mi.tmp.d()};
    ew_lincomb(mi.A.d(), mi.rows * mi.R, 1, c, x, nullptr, stream_);
  }
  SysBuild sb;
  sb.ngram = 0;
  sb.Cpre = b.Csys.d();
  sb.w = t.weight;
  sb.ridge = has_ridge_ ? mi.ridge : 0.0;
  sb.bsum_half = opt.bsum ? opt.bsum_weight / 2 : 0.0;
  sb.rho_scale = 1.0;
  sb.nrho = nrho;
  sb.R = mi.R;
  sb.C = mi.C.d(); sb.rho = mi.rho.d(); sb.Bsys = mi.Bsys.d(); sb.L = mi.L.d();
  sb.Binv = nrho > 0 ? mi.Binv.d() : nullptr;
  sb.ctl = ctl_of_mode(m);
  sys_build(sb, stream_);
  t.last_pos = 0;                                     // last_m(p) = 1  (:168)
  mi.Aeff = mi.A.d();
  if (opt.bsum) {
    Coef c[2] = {coef(1.0), coef(opt.bsum_weight / 2)};
    const double* x[2] = {mi.A.d(), mi.fac.d()};
    ew_lincomb(mi.Ab.d(), mi.rows * mi.R, 2, c, x, nullptr, stream_);
    mi.Aeff = mi.Ab.d();
  }
}

// mode B (:191-218) + ADMM_B_Parafac2 (:509-589)
void Engine::par2_update_B(int m, const aoadmm_options& opt, int iter) {
  ModeInfo& mi = modes_[m];
  TensorInfo& t = tensors_[mi.tensor];
  Par2Block& b = t.p2;
  const P2Dims d = b.dims();
  ModeInfo& mA = modes_[t.modes[0]];
  ModeInfo& mC = modes_[t.modes[2]];
  AdmmCtl* ctl = ctl_of_mode(m);
  const bool constr = mi.constrained && iter >= opt.iter_start_PAR2Bkconstraint;        // :209, :527
  par2_xta(b.X.d(), mA.fac.d(), mC.fac.d(), t.weight, d, b.Ak.d(), stream_);
  par2_b_system(mA.gram.d(), mC.fac.d(), t.weight, has_ridge_ ? mi.ridge : 0.0, opt.bsum ? opt.bsum_weight / 2 : 0.0,
                opt.has_increase_factor_rhoBk ? opt.increase_factor_rhoBk : 1.0, 1 + (constr ? 1 : 0), d, b.rhok.d(),
                b.Lk.d(), ctl, stream_);
  if (opt.bsum) {                                                                        // :204-207
    Coef c[2] = {coef(1.0), coef(opt.bsum_weight / 2)};
    const double* x[2] = {b.Ak.d(), mi.fac.d()};
    ew_lincomb(b.Ak.d(), b.Jtot * b.R, 2, c, x, nullptr, stream_);
  }
  t.last_pos = 1;                                                                        // last_m(p) = 2; par2_b_system opened the loop (ctl)
  P2BArgs a;
  a.Ak = b.Ak.d(); a.L = b.Lk.d(); a.rho = b.rhok.d();
  a.B = mi.fac.d(); a.P = b.P.d(); a.Pold = b.Pold.d(); a.mu = b.muDB.d(); a.W = b.W.d();
  a.DeltaB = b.DeltaB.d(); a.DeltaBold = b.DeltaBold.d(); a.part = b.part.d();
  a.Z = constr ? mi.Z.d() : nullptr; a.muZ = constr ? mi.mu.d() : nullptr;
  a.norms = b.norms.d();
  a.use_constr = constr ? 1 : 0;
  b.Jrot.ensure((size_t)b.K * b.R * b.R * sizeof(double));
  a.Jrot = b.Jrot.d();
  const P2AllReduce ar = [this](double* buf, int64_t n) { allreduce(buf, n); };
  double* psum = b.slab_sharded ? b.psum.d() : nullptr;
  double* part4 = b.slab_sharded ? b.psum.d() + (int64_t)b.R * b.R + 8 : nullptr;
  if (par2_b_loop_folded_ok(d, constr, b.slab_sharded)) {
    par2_b_loop_folded(a, d, ctl, opt.MaxInnerIters, opt.innerRelPrTol_coupl, opt.innerRelPrTol_constr,
                       opt.innerRelDualTol_coupl, opt.innerRelDualTol_constr, stream_);
    par2_gram(mi.fac.d(), d, b.GB.d(), stream_);                                         // :216-218
    mi.version++;
    return;
  }
  for (int it = 0; it < opt.MaxInnerIters; ++it) {
    par2_b_iteration(a, d, ctl, stream_, psum, ar);
    if (constr)
      par2_b_constraint(mi.prox, mi.fac.d(), mi.Z.d(), mi.mu.d(), mi.Zold.d(), mi.V.d(), b.rhok.d(), d, mi.proxws.d(),
                        b.norms.d(), ctl, stream_);
    par2_b_finalize(b.norms.d(), d, constr ? 1 : 0, ctl, opt.MaxInnerIters, opt.innerRelPrTol_coupl,
                    opt.innerRelPrTol_constr, opt.innerRelDualTol_coupl, opt.innerRelDualTol_constr, stream_, part4, ar);
  }
  par2_gram(mi.fac.d(), d, b.GB.d(), stream_);                                           // :216-218
  mi.version++;
}

// mode C, uncoupled (:219-248; constrained rows through ADMM_constrained_only :602-606)
void Engine::par2_update_C(int m, const aoadmm_options& opt) {
  ModeInfo& mi = modes_[m];
  TensorInfo& t = tensors_[mi.tensor];
  Par2Block& b = t.p2;
  const P2Dims d = b.dims();
  ModeInfo& mA = modes_[t.modes[0]];
  ModeInfo& mB = modes_[t.modes[1]];
  AdmmCtl* ctl = ctl_of_mode(m);
  par2_xkb(b.X.d(), mB.fac.d(), d, b.T1.d(), stream_);
  const size_t RR = (size_t)b.R * b.R;
  if (b.slab_sharded) {                               // rows of the other ranks arrive through the all-reduce
    AO_HIP(hipMemsetAsync(b.ac.p, 0, (size_t)b.K * b.R * 8, stream_));
    AO_HIP(hipMemsetAsync(b.rhoc.p, 0, (size_t)b.K * 8, stream_));
    AO_HIP(hipMemsetAsync(b.Lc.p, 0, (size_t)b.K * RR * 8, stream_));
  }
  par2_c_system(mA.fac.d(), b.T1.d(), mA.gram.d(), b.GB.d(), t.weight, has_ridge_ ? mi.ridge : 0.0,
                opt.bsum ? opt.bsum_weight / 2 : 0.0, mi.constrained ? 1 : 0, 0, d, mi.fac.d(), b.ac.d(), b.rhoc.d(),
                b.Lc.d(), ctl, stream_);
  if (b.slab_sharded) {
    allreduce(b.ac.d(), (int64_t)b.K * b.R);
    allreduce(b.rhoc.d(), b.K);
    allreduce(b.Lc.d(), (int64_t)b.K * RR);
  }
  const bool wg_loop = mi.constrained && admm_loop_wg_ok(mi.rows, mi.R, mi.prox.type, opt.MaxInnerIters);
  if (!wg_loop) par2_rho_max(b.rhoc.d(), b.K, b.rhomax.d(), stream_);   // the one-workgroup loop takes max(rho) itself
  const P2Dims dall = b.dims_all();                   // the K x R row systems are solved on every rank
  t.last_pos = 2;                                                                        // last_m(p) = 3
  if (wg_loop) {
    // K <= 256 rows: the K row systems, update_constraint with max(rho) and the loop test in one launch (:602-606)
    WgLoopU wa;
    wa.A = b.ac.d(); wa.Binv = nullptr; wa.L = b.Lc.d(); wa.rho = b.rhoc.d(); wa.rho_prox = nullptr;
    wa.fac = mi.fac.d(); wa.Z = mi.Z.d(); wa.mu = mi.mu.d();
    wa.rows = mi.rows; wa.R = mi.R; wa.per_row = 1;
    wa.ptype = mi.prox.type; wa.p0 = mi.prox.p0; wa.p1 = mi.prox.p1;
    wa.max_inner = opt.MaxInnerIters; wa.tol_pr = opt.innerRelPrTol_constr; wa.tol_du = opt.innerRelDualTol_constr;
    wa.ctl = ctl; wa.reset = 1;
    wa.gram = nullptr; wa.facT = nullptr;
    admm_loop_wg(wa, stream_);
    mi.version++;
    return;
  }
  ctl_reset(ctl, stream_);
  if (!mi.constrained) {
    par2_c_rowsolve(b.ac.d(), b.rhoc.d(), b.Lc.d(), nullptr, nullptr, 0, dall, mi.fac.d(), nullptr, stream_);   // :236
  } else {
    double* sl = resid_slots(m);
    FinalizeArgs fa;
    fa.nmodes = 1; fa.max_inner = opt.MaxInnerIters;
    fa.tol_pr_coupl = fa.tol_du_coupl = 1e300;          // only the constraint residuals steer this loop (:600)
    fa.tol_pr_constr = opt.innerRelPrTol_constr; fa.tol_du_constr = opt.innerRelDualTol_constr;
    fa.slots[0] = sl; fa.constrained[0] = 1; fa.coupled[0] = 0;
    for (int it = 0; it < opt.MaxInnerIters; ++it) {
      par2_c_rowsolve(b.ac.d(), b.rhoc.d(), b.Lc.d(), mi.Z.d(), mi.mu.d(), 1, dall, mi.fac.d(), ctl, stream_);  // :603-606
      // update_constraint with max(rho) (:1423-1424)
      constraint_update(mi.prox, mi.fac.d(), mi.Z.d(), mi.mu.d(), mi.Zold.d(), mi.V.d(), mi.rows, mi.R, b.rhomax.d(),
                        1.0, mi.proxws.d(), sl, redws_.d(), ctl, stream_);
      admm_finalize_generic(fa, ctl, stream_);
    }
  }
  mi.version++;
}

// mode C inside a coupling (:253-385): right-hand sides a_k, rho_k and the systems the coupled ADMM needs --
// types 0, 3, 4: per-row Cholesky factors of w*C_k + rho_k/2*I (+ rho_k/2*I if constrained) (:260-267, :327-334, :349-356);
// type 2: the same with rho_k/2*H*H' for the coupling term (:305-312);
// types 1, 5: the (K*R) x (K*R) system blkdiag(w*C_k) + rhoC/2*kron(H'H, I) (+ rhoC/2*I), rhoC = mean(rho) (:282-297, :371-385)
void Engine::par2_prepare_C_coupled(int m, int ctype, const aoadmm_options& opt) {
  ModeInfo& mi = modes_[m];
  TensorInfo& t = tensors_[mi.tensor];
  Par2Block& b = t.p2;
  const P2Dims d = b.dims();
  ModeInfo& mA = modes_[t.modes[0]];
  ModeInfo& mB = modes_[t.modes[1]];
  AdmmCtl* ctl = ctl_of_mode(m);
  const int con = mi.constrained ? 1 : 0;
  const bool big = ctype == 1 || ctype == 5;          // H*C = ...: one (K*R) x (K*R) system instead of K row systems
  b.rhosum.ensure(64);
  par2_xkb(b.X.d(), mB.fac.d(), d, b.T1.d(), stream_);
  par2_c_system(mA.fac.d(), b.T1.d(), mA.gram.d(), b.GB.d(), t.weight, has_ridge_ ? mi.ridge : 0.0,
                opt.bsum ? opt.bsum_weight / 2 : 0.0, big ? 0 : (ctype == 2 ? con : 1 + con), big ? 1 : 0, d,
                mi.fac.d(), b.ac.d(), b.rhoc.d(), b.Lc.d(), ctl, stream_, ctype == 2 ? mi.HHt.d() : nullptr);
  // max(rho) for the prox (:1424); mean(rho) takes the place of the scalar rho of a CP mode (:284, :712), sum(rho)
  // weighs this mode in the Delta update (:736)
  par2_rho_max(b.rhoc.d(), b.K, b.rhomax.d(), stream_, mi.rho.d(), b.rhosum.d());
  t.last_pos = 2;                                                              // last_m(p) = 3
  mi.Aeff = b.ac.d();
  if (big) {
    const int n = b.K * b.R;
    AO_REQUIRE(mi.hc == b.K, "coupling matrix of the PARAFAC2 C mode has %lld columns, the mode has %d rows", (long long)mi.hc, b.K);
    if (!b.have_HtH) {
      // H'H once per model, on the host: when it is diagonal (H selects or scales rows -- every sampling-rate
      // coupling, script 14) kron(H'H, I) is diagonal too and the (K*R)-system falls apart into K row systems
      const int64_t hr = mi.hr;
      std::vector<double> hth((size_t)b.K * b.K, 0.0);
      bool diag = true;
      for (int a = 0; a < b.K; ++a)
        for (int c2 = 0; c2 < b.K; ++c2) {
          double acc = 0.0;
          for (int64_t q = 0; q < hr; ++q) acc += mi.H_host[(size_t)q + (size_t)hr * a] * mi.H_host[(size_t)q + (size_t)hr * c2];
          hth[(size_t)a + (size_t)b.K * c2] = acc;
          if (a != c2 && acc != 0.0) diag = false;
        }
      b.hth_diag = diag;
      if (diag) {
        std::vector<double> dd(b.K);
        for (int a = 0; a < b.K; ++a) dd[a] = hth[(size_t)a + (size_t)b.K * a];
        b.HtH.ensure((size_t)b.K * 8);
        AO_HIP(hipMemcpyAsync(b.HtH.p, dd.data(), (size_t)b.K * 8, hipMemcpyHostToDevice, stream_));
      } else {
        b.HtH.ensure((size_t)b.K * b.K * 8);
        AO_HIP(hipMemcpyAsync(b.HtH.p, hth.data(), hth.size() * 8, hipMemcpyHostToDevice, stream_));
      }
      AO_HIP(hipStreamSynchronize(stream_));          // hth / dd are locals
      b.have_HtH = true;
    }
    if (b.hth_diag) {
      // row k: C(k,:) * (B_k + rhoC/2*(d_k [+ 1])*I) = A_inner(k,:); b.Lc holds B_k and is overwritten by the factors
      par2_c_rowsys_diag(b.Lc.d(), b.HtH.d(), mi.rho.d(), con, b.K, b.R, ctl, stream_);
    } else {
      AO_REQUIRE(n <= kDenseMaxN, "PARAFAC2 C mode coupled with type %d through a matrix whose H'H is not diagonal: K*R = %d exceeds the dense-system limit %d", ctype, n, kDenseMaxN);
      b.Mbig.ensure((size_t)n * n * 8); b.Minv.ensure((size_t)n * n * 8);
      par2_c_big_system(b.Lc.d(), b.HtH.d(), mi.rho.d(), con, b.K, b.R, b.Mbig.d(), stream_);
      dense_spd_inverse(b.Mbig.d(), b.Minv.d(), n, ctl, stream_);
    }
  }
}

// objective pieces of a PARAFAC2 block that do not go through the last_mttkrp shortcut (:1262-1264, :1355, :1337)
void Engine::par2_objective_enqueue(TensorInfo& t) {
  Par2Block& b = t.p2;
  const P2Dims d = b.dims();
  ModeInfo& mA = modes_[t.modes[0]];
  ModeInfo& mB = modes_[t.modes[1]];
  ModeInfo& mC = modes_[t.modes[2]];
  const bool regs = mB.constrained && prox_has_reg_value(mB.prox.type);
  if (b.slab_sharded) {                               // per-slab values of the other ranks arrive through the all-reduce
    AO_HIP(hipMemsetAsync(b.res.p, 0, (size_t)b.K * 8, stream_));
    AO_HIP(hipMemsetAsync(b.q.p, 0, (size_t)b.K * 4 * 8, stream_));
    if (regs) AO_HIP(hipMemsetAsync(b.regv.p, 0, (size_t)b.K * 8, stream_));
  }
  par2_residual(b.X.d(), mA.fac.d(), mB.fac.d(), mC.fac.d(), d, b.res.d(), stream_);
  par2_b_gaps(mB.fac.d(), b.P.d(), b.DeltaB.d(), mB.constrained ? mB.Z.d() : nullptr, d, b.q.d(), stream_);
  if (regs) par2_reg_values(mB.fac.d(), mB.prox.type, mB.prox.p0, d, b.regv.d(), stream_);
  if (b.slab_sharded) {
    // res[K] carries the not-positive-definite flags of this block's modes, so that a Cholesky failure in one
    // rank's slabs stops every rank (finish_eval) instead of leaving the others waiting in a collective
    par2_collect_notpd(ctl_of_mode(t.modes[0]), ctl_of_mode(t.modes[1]), ctl_of_mode(t.modes[2]), b.res.d() + b.K, stream_);
    allreduce(b.res.d(), b.K + 1);
    allreduce(b.q.d(), (int64_t)b.K * 4);
    if (regs) allreduce(b.regv.d(), b.K);
  }
}

// End of a solve with slab-sharded blocks: every rank receives the slabs the others updated (G.fac{B}, G.P,
// G.mu_DeltaB and, if B_k is constrained, its split and dual variables), so aoadmm_state_get returns the same
// struct on every rank.
void Engine::par2_gather_slabs(TensorInfo& t) {
  Par2Block& b = t.p2;
  if (!b.slab_sharded) return;
  ModeInfo& mB = modes_[t.modes[1]];
  const int64_t n = b.Jtot * b.R, e0 = b.off_h[b.k0] * b.R, e1 = b.off_h[b.k1] * b.R;
  auto gather = [&](DevBuf& buf) {
    if (e0 > 0) AO_HIP(hipMemsetAsync(buf.p, 0, (size_t)e0 * 8, stream_));
    if (e1 < n) AO_HIP(hipMemsetAsync(buf.d() + e1, 0, (size_t)(n - e1) * 8, stream_));
    allreduce(buf.d(), n);
  };
  gather(mB.fac); gather(b.P); gather(b.muDB);
  if (mB.constrained && mB.has_Z) { gather(mB.Z); gather(mB.mu); }
  mB.version++;
}

}  // namespace aoadmm
