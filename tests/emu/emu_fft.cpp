// TEST-ONLY host emulation of the libniftyk transform kernels (see nk_core.h).  Runs the very same
// phase functions the GPU kernels run, for tid = 0..nthreads-1 sequentially, on host memory.  Used by
// tests/test_kernel_emulation.py to validate index math / twiddles / fusion on a CPU-only box.
// Never linked into the product library and never used as a fallback.
#define NK_HOST_EMU 1
#include <cstring>
#include <cstdlib>
#include <vector>

#include "../../nifty_amd/csrc/nk_plan.h"
#include "../../nifty_amd/csrc/nk_fft2.h"

template <typename T>
static std::vector<C2<T>> conv_tw(const std::vector<double>& tw) {
  std::vector<C2<T>> r(tw.size() / 2);
  for (size_t i = 0; i < r.size(); ++i) r[i] = C2<T>{(T)tw[2 * i], (T)tw[2 * i + 1]};
  return r;
}

template <typename T>
static void run_stages(C2<T>* lds, int nthr, const NkLinePlan& lp, const NkTile& tl, const C2<T>* tw) {
  int L = lp.n;
  for (int s = 0; s < lp.nstage; ++s) {
    const int R = lp.radix[s];
    for (int tid = 0; tid < nthr; ++tid) {
      NK_STAGE_DISPATCH(R, lds, tid, nthr, lp, tl, L, tw, s)
    }
    L /= R;
  }
}

template <typename T>
static int emu_run(int ndim, const int64_t* shape, int dtype, int64_t batch, const nk_fuse* f, int convention) {
  NkHostPlan hp;
  const char* msg;
  int rc = nk_host_plan_init(hp, ndim, shape, dtype, batch, &msg);
  if (rc != NK_OK) return rc;
  auto tw_a = conv_tw<T>(hp.tw_a), twr = conv_tw<T>(hp.twr_a), tw_b = conv_tw<T>(hp.tw_b), tw_c = conv_tw<T>(hp.tw_c);
  NkPassA pa = hp.pa;
  pa.g.sign = convention == NK_HARTLEY_CANONICAL ? -1 : 1;
  double energy = 0.0;
  std::vector<C2<T>> lds(hp.lds_a / sizeof(C2<T>) + 16);
  const int64_t blocks_a = (pa.nlines + pa.tl.tile - 1) / pa.tl.tile;
  std::vector<C2<T>> work(hp.work_bytes / sizeof(C2<T>) + 1), scratch(hp.scratch_bytes / sizeof(C2<T>) + 1);
  for (int64_t blk = 0; blk < blocks_a; ++blk) {
    const int nthr = hp.threads_a;
    for (int tid = 0; tid < nthr; ++tid) nk_passA_load<T>(pa, *f, blk, tid, nthr, lds.data());
    run_stages<T>(lds.data(), nthr, pa.lp, pa.tl, tw_a.data());
    for (int tid = 0; tid < nthr; ++tid) {
      if (ndim == 1) nk_pass1d_store<T>(pa, *f, blk, tid, nthr, lds.data(), twr.data(), energy);
      else nk_passA_store<T>(pa, blk, tid, nthr, lds.data(), twr.data(), work.data());
    }
  }
  if (ndim == 3) {
    lds.assign(hp.lds_b / sizeof(C2<T>) + 16, C2<T>{0, 0});
    const int64_t blocks = hp.pb.outer * hp.pb.tiles_per_slab;
    for (int64_t blk = 0; blk < blocks; ++blk) {
      const int nthr = hp.threads_b;
      for (int tid = 0; tid < nthr; ++tid) nk_passS_load<T>(hp.pb, blk, tid, nthr, lds.data(), work.data());
      run_stages<T>(lds.data(), nthr, hp.pb.lp, hp.pb.tl, tw_b.data());
      for (int tid = 0; tid < nthr; ++tid) nk_passB_store<T>(hp.pb, blk, tid, nthr, lds.data(), work.data());
    }
  }
  if (ndim >= 2) {
    NkPassS pc = hp.pc;
    pc.g.sign = pa.g.sign;
    lds.assign(hp.lds_c / sizeof(C2<T>) + 16, C2<T>{0, 0});
    const int64_t blocks = pc.outer * pc.tiles_per_slab;
    for (int64_t blk = 0; blk < blocks; ++blk) {
      const int nthr = hp.threads_c;
      for (int tid = 0; tid < nthr; ++tid) nk_passS_load<T>(pc, blk, tid, nthr, lds.data(), work.data());
      run_stages<T>(lds.data(), nthr, pc.lp, pc.tl, tw_c.data());
      for (int tid = 0; tid < nthr; ++tid)
        nk_passC_store<T>(pc, *f, blk, tid, nthr, lds.data(), scratch.data(), energy);
    }
    const int64_t total_d = (int64_t)pc.g.batch * pc.g.nm * pc.g.na;
    for (int64_t gid = 0; gid < total_d; ++gid) nk_passD<T>(pc.g, *f, gid, scratch.data(), energy);
  }
  if ((f->epi == NK_EPI_LIKELIHOOD || f->epi == NK_EPI_VJP) && f->value) *f->value += energy;
  return NK_OK;
}

// ---- fast path (register-resident bodies of nk_fft2.h) ------------------------------------------------
template <typename T, int E>
struct HostExec {
  std::vector<PassRegs<T, E>> regs;
  explicit HostExec(int n) : regs(n) {}
  template <typename F>
  void phase(F f) {
    for (int tid = 0; tid < (int)regs.size(); ++tid) f(tid, regs[tid]);
  }
  template <typename F>
  void last_phase(F f) {
    phase(f);
  }
};

template <typename T, int N, int MODE>
static void emu2_strided_m(NkPassS p, const nk_fuse& f, const C2<T>* tw, C2<T>* work, C2<T>* scratch, double* energy) {
  using ST = StridedTile<T, N, false, MODE>;
  p.tl.tile = ST::TILE;
  p.tl.dtile = nk_make_div(ST::TILE);
  p.tiles_per_slab = (int)(p.inner / ST::TILE);
  std::vector<T> plane(StridedTile<T, N, true>::LDS_BYTES / sizeof(T));  // room for the complex-plane classes
  const int64_t blocks = p.outer * p.tiles_per_slab;
  for (int64_t blk = 0; blk < blocks; ++blk) {
    HostExec<T, ST::SC::E> ex(ST::THREADS);
    if constexpr (MODE == 3) {
      if (sizeof(T) == 8 && f.field_octant && f.pro == NK_PRO_AMP && f.io32) nk_strided_body<T, N, ST::TILE, 3, 9>(ex, p, f, blk, plane.data(), tw, work, scratch, energy);
      else if (f.field_octant && f.pro == NK_PRO_AMP) nk_strided_body<T, N, ST::TILE, 3, 4>(ex, p, f, blk, plane.data(), tw, work, scratch, energy);
      else if (f.field_octant && f.pro == NK_PRO_AMP_JVP) nk_strided_body<T, N, ST::TILE, 3, 5>(ex, p, f, blk, plane.data(), tw, work, scratch, energy);
      else if (f.pro == NK_PRO_PLAIN) nk_strided_body<T, N, ST::TILE, 3, 0, true>(ex, p, f, blk, plane.data(), tw, work, scratch, energy);
      else if (f.pro == NK_PRO_MUL) nk_strided_body<T, N, ST::TILE, 3, 6, true>(ex, p, f, blk, plane.data(), tw, work, scratch, energy);
      else if (f.pro == NK_PRO_AMP && f.afield) nk_strided_body<T, N, ST::TILE, 3, 1, true>(ex, p, f, blk, plane.data(), tw, work, scratch, energy);
      else if (f.pro == NK_PRO_AMP_JVP && f.afield && f.dafield) nk_strided_body<T, N, ST::TILE, 3, 3>(ex, p, f, blk, plane.data(), tw, work, scratch, energy);
      else if (f.pro == NK_PRO_AMP_JVP && f.afield && f.dampT) nk_strided_body<T, N, ST::TILE, 3, 2>(ex, p, f, blk, plane.data(), tw, work, scratch, energy);
      else nk_strided_body<T, N, ST::TILE, 3, -1>(ex, p, f, blk, plane.data(), tw, work, scratch, energy);
    } else {
      nk_strided_body<T, N, ST::TILE, 0, -1>(ex, p, f, blk, plane.data(), tw, work, scratch, energy);
    }
  }
}
// one launch of the two-level first-axis pass (k2_tl of nk_fft_t.hip): MODE 4 with the prologue classes of nk_tl_first, MODE 5
template <typename T, int N, int MODE>
static void emu_tl(NkPassS p, int other, const nk_fuse& f, const C2<T>* tw, const C2<T>* tw_full, C2<T>* work) {
  using ST = StridedTile<T, N, false, MODE>;
  p.tl.tile = ST::TILE;
  p.tl.dtile = nk_make_div(ST::TILE);
  p.tiles_per_slab = (int)(p.inner / ST::TILE);
  p.sub = other;
  std::vector<T> plane(ST::LDS_BYTES / sizeof(T));
  double energy = 0.0;
  const int64_t blocks = (int64_t)p.g.batch * other * p.tiles_per_slab;
  for (int64_t blk = 0; blk < blocks; ++blk) {
    HostExec<T, ST::SC::E> ex(ST::THREADS);
    const int64_t b = nk_xcd_contig(blk, blocks);
#define NK_TL(PC) nk_strided_body<T, N, ST::TILE, MODE, PC>(ex, p, f, b, plane.data(), tw, work, (C2<T>*)nullptr, &energy, nullptr, tw_full)
    if constexpr (MODE == 4) {
      if (sizeof(T) == 8 && f.field_octant && f.pro == NK_PRO_AMP && f.io32) NK_TL(9);
      else if (f.field_octant && f.pro == NK_PRO_AMP) NK_TL(4);
      else if (f.field_octant && f.pro == NK_PRO_AMP_JVP) NK_TL(5);
      else if (f.pro == NK_PRO_PLAIN) NK_TL(0);
      else if (f.pro == NK_PRO_MUL) NK_TL(6);
      else NK_TL(-1);
    } else {
      NK_TL(-1);
    }
#undef NK_TL
  }
}
template <typename T, int N>
static void emu2_strided(NkPassS p, int mode, const nk_fuse& f, const C2<T>* tw, C2<T>* work, C2<T>* scratch,
                         double* energy) {
  if (mode == 3) emu2_strided_m<T, N, 3>(p, f, tw, work, scratch, energy);
  else emu2_strided_m<T, N, 0>(p, f, tw, work, scratch, energy);
}

template <typename T, int H>
static void emu2_contig(NkPassA p, bool is_1d, const nk_fuse& f, const C2<T>* tw, const C2<T>* twr, C2<T>* work,
                        double* energy) {
  using CT = ContigTile<T, H>;
  std::vector<T> planes(CT::LDS_BYTES / sizeof(T));
  const int64_t blocks = (p.nlines + CT::TILE - 1) / CT::TILE;
  for (int64_t blk = 0; blk < blocks; ++blk) {
    HostExec<T, Sched<T, H>::E> ex(CT::THREADS);
    if (is_1d) nk_contig_body<T, H, CT::TILE, true>(ex, p, f, blk, planes.data(), tw, twr, work, energy);
    else nk_contig_body<T, H, CT::TILE, false>(ex, p, f, blk, planes.data(), tw, twr, work, energy);
  }
}

template <typename T, int NL, bool COUPLES, int EC, int PAIR = 0>
static void emu2_final_ec(const NkPassF& pf0, const nk_fuse& f, const C2<T>* tw, const C2<T>* work, double* energy) {
  using CT = FinalTile<T, NL, EC, COUPLES ? 2 : 1>;
  NkPassF pf = pf0;
  pf.tiles_per_a = (COUPLES && pf.A > 1 && CT::TILE >= 2) ? (pf.M / 2 + 1 + CT::TILE / 2 - 1) / (CT::TILE / 2)
                                                          : (pf.M + CT::TILE - 1) / CT::TILE;
  std::vector<T> planes(CT::LDS_BYTES / sizeof(T));
  const int64_t blocks = (int64_t)pf.g.batch * (pf.A / 2 + 1) * pf.tiles_per_a;
  for (int64_t blk = 0; blk < blocks; ++blk) {
    HostExec<T, SchedF<T, NL>::E> ex(CT::THREADS);
    nk_final_body<T, NL, CT::TILE, COUPLES, EC, PAIR>(ex, pf, f, blk, planes.data(), tw, work, energy);
  }
}

template <typename T, int NL>
static void emu2_final(const NkPassF& pf, const nk_fuse& f, const C2<T>* tw, const C2<T>* work, double* energy) {
  const bool couples = f.epi == NK_EPI_VJP;  // same dispatch as nk_launch_final
  if (couples && f.afield && nk_final_single_2d<T, NL>() && pf.A == 1) emu2_final_ec<T, NL, false, 2>(pf, f, tw, work, energy);  // as nk_launch_final
  else if (couples && f.afield) emu2_final_ec<T, NL, true, 2>(pf, f, tw, work, energy);
  else if (couples) emu2_final_ec<T, NL, true, -1>(pf, f, tw, work, energy);
  else if (f.epi == NK_EPI_AFFINE) emu2_final_ec<T, NL, false, 0>(pf, f, tw, work, energy);
  else if (f.epi == NK_EPI_MUL) emu2_final_ec<T, NL, false, 1>(pf, f, tw, work, energy);
  else if (f.epi == NK_EPI_LIKELIHOOD && sizeof(T) == 8 && f.io32) emu2_final_ec<T, NL, false, 5>(pf, f, tw, work, energy);
  else if (f.epi == NK_EPI_LIKELIHOOD) emu2_final_ec<T, NL, false, 3>(pf, f, tw, work, energy);
  else emu2_final_ec<T, NL, false, -1>(pf, f, tw, work, energy);
}

// ---- sandwich pipeline (nk_fft3.h) ----------------------------------------------------------------------
template <typename T, int NL>
static void emu3_final(const NkPassF& pf, const nk_fuse& f, const C2<T>* tw, const C2<T>* work, double* energy) {
  if (f.epi == NK_EPI_VJP && f.afield) emu2_final_ec<T, NL, true, 2, 1>(pf, f, tw, work, energy);  // as nk_launch_final3
  else if (f.epi == NK_EPI_VJP) emu2_final_ec<T, NL, true, -1, 1>(pf, f, tw, work, energy);
  else if (f.epi == NK_EPI_AFFINE) emu2_final_ec<T, NL, false, 0, 1>(pf, f, tw, work, energy);
  else emu2_final_ec<T, NL, false, -1, 1>(pf, f, tw, work, energy);
}

template <typename T, int H>
static void emu3_contig(const NkPass3& p, const nk_fuse& f, const C2<T>* tw, const C2<T>* twr, C2<T>* work) {
  using CT = Contig3Tile<T, H>;
  // QUAD launches of the octant classes on 3-D grids: as nk_launch_contig3 (NK_CONTIG_QUAD=0 switches them off there too)
  const char* qenv = getenv("NK_CONTIG_QUAD");
  if constexpr (CT::QUAD_OK) {
    if (f.field_octant && p.g.ndim == 3 && (f.pro == NK_PRO_AMP || f.pro == NK_PRO_AMP_JVP) && !(qenv && atoi(qenv) == 0)) {
      std::vector<T> qplanes(CT::QLDS_BYTES / sizeof(T));
      // the launch grid of nk_launch_contig3: x = the (a8, b8) index inside a batch member, y = the member
      const int64_t batch = p.nlines / ((int64_t)p.g.na * p.g.nm);
      const int64_t per = (int64_t)(p.g.na / 2 + 1) * (p.g.nm / 2 + 1);
      NkPass3 pq = p;
      pq.dmh = nk_make_div(p.g.nm / 2 + 1);
      for (int bat = 0; bat < (int)batch; ++bat)
        for (int64_t blk = 0; blk < per; ++blk) {
          HostExec<T, CT::SC::E> ex(CT::QTHREADS);
          if (f.pro == NK_PRO_AMP) nk_contig3_body<T, H, 4, 4, true>(ex, pq, f, blk, qplanes.data(), tw, twr, work, bat);
          else if (f.cg_r && f.dafield) nk_contig3_body<T, H, 4, 8, true>(ex, pq, f, blk, qplanes.data(), tw, twr, work, bat);
          else if (f.pidx_octant && f.dampT) nk_contig3_body<T, H, 4, 7, true>(ex, pq, f, blk, qplanes.data(), tw, twr, work, bat);
          else nk_contig3_body<T, H, 4, 5, true>(ex, pq, f, blk, qplanes.data(), tw, twr, work, bat);
        }
      return;
    }
  }
  std::vector<T> planes(CT::LDS_BYTES / sizeof(T));
  const int64_t blocks = (p.nlines + CT::TILE - 1) / CT::TILE;
  for (int64_t blk = 0; blk < blocks; ++blk) {
    HostExec<T, CT::SC::E> ex(CT::THREADS);
    if (f.field_octant && f.pro == NK_PRO_AMP) nk_contig3_body<T, H, CT::TILE, 4>(ex, p, f, blk, planes.data(), tw, twr, work);
    else if (f.field_octant && f.pro == NK_PRO_AMP_JVP && f.cg_r && f.dafield) nk_contig3_body<T, H, CT::TILE, 8>(ex, p, f, blk, planes.data(), tw, twr, work);
    else if (f.field_octant && f.pro == NK_PRO_AMP_JVP && f.pidx_octant && f.dampT) nk_contig3_body<T, H, CT::TILE, 7>(ex, p, f, blk, planes.data(), tw, twr, work);
    else if (f.field_octant && f.pro == NK_PRO_AMP_JVP) nk_contig3_body<T, H, CT::TILE, 5>(ex, p, f, blk, planes.data(), tw, twr, work);
    else if (f.pro == NK_PRO_PLAIN) nk_contig3_body<T, H, CT::TILE, 0>(ex, p, f, blk, planes.data(), tw, twr, work);
    else if (f.pro == NK_PRO_MUL) nk_contig3_body<T, H, CT::TILE, 6>(ex, p, f, blk, planes.data(), tw, twr, work);
    else nk_contig3_body<T, H, CT::TILE, -1>(ex, p, f, blk, planes.data(), tw, twr, work);
  }
}

template <typename Regs>
struct HostExecR {
  std::vector<Regs> regs;
  explicit HostExecR(int n) : regs(n) {}
  template <typename F>
  void phase(F f) {
    for (int tid = 0; tid < (int)regs.size(); ++tid) f(tid, regs[tid]);
  }
  template <typename F>
  void last_phase(F f) {
    phase(f);
  }
};

// mode bit 0: complex-plane exchange, bit 1: persistent launch with register prefetch (5 workgroups), bit 2: wide
// schedule (SchedW) with the half-column complex exchange
template <typename T, int N>
static void emu3_mid(NkPassM pm, const nk_fuse& f, const C2<T>* tw, C2<T>* work, int mode) {
  const bool cx = mode & 1, pf = mode & 2, wide = (mode & 4) && SchedW<T, N>::E != Sched<T, N>::E, twc = mode & 8;
  using STN = StridedTile<T, N, true>;
  using STW = StridedTile<T, N, false, 0>;
  const int tile = wide ? STW::TILE : STN::TILE, threads = wide ? STW::THREADS : STN::THREADS;
  pm.s.tl.tile = tile;
  pm.s.tl.dtile = nk_make_div(tile);
  pm.s.tiles_per_slab = (int)(pm.s.inner / tile);
  std::vector<T> plane(N * STN::TILE * 2 + 16);
  const int64_t blocks = pm.s.outer * pm.s.tiles_per_slab;
  const int64_t nwg = pf ? (5 < blocks ? 5 : blocks) : blocks;
  for (int64_t wg = 0; wg < nwg; ++wg) {
#define NK_MID(XMV, MFV, PFV, SCV, TILEV)                                                                      \
  {                                                                                                            \
    HostExecR<MidRegs<T, SCV::E, PFV>> ex(threads);                                                                 \
    nk_mid_body<T, N, TILEV, XMV, MFV, PFV, SCV>(ex, pm, f, wg, nwg, blocks, 1, plane.data(), tw, work);      \
  }
#define NK_MID_MF(XMV, PFV, SCV, TILEV)                   \
  if (f.mul) NK_MID(XMV, true, PFV, SCV, TILEV) else NK_MID(XMV, false, PFV, SCV, TILEV)
    using SN = Sched<T, N>;
    using SW = SchedW<T, N>;
    if (wide) {
      if (pf) NK_MID_MF(2, true, SW, STW::TILE) else NK_MID_MF(2, false, SW, STW::TILE)
    } else if (cx) {
      if (pf) NK_MID_MF(1, true, SN, STN::TILE) else NK_MID_MF(1, false, SN, STN::TILE)
    } else if (twc) {  // composed twiddles (the two-workgroups-per-CU configuration of k3_mid)
      HostExecR<MidRegs<T, SN::E, false>> ex(threads);
      if (f.mul) nk_mid_body<T, N, STN::TILE, 0, true, false, SN, true>(ex, pm, f, wg, nwg, blocks, 1, plane.data(), tw, work);
      else nk_mid_body<T, N, STN::TILE, 0, false, false, SN, true>(ex, pm, f, wg, nwg, blocks, 1, plane.data(), tw, work);
    } else {
      if (pf) NK_MID_MF(0, true, SN, STN::TILE) else NK_MID_MF(0, false, SN, STN::TILE)
    }
#undef NK_MID_MF
#undef NK_MID
  }
}

template <typename T>
static int emu4_run(int ndim, const int64_t* shape, int dtype, int64_t batch, const nk_fuse* f, double scale_first,
                    int convention, int cx) {
  NkHostPlan hp;
  const char* msg;
  int rc = nk_host_plan_init(hp, ndim, shape, dtype, batch, &msg);
  if (rc != NK_OK) return rc;
  const NkGeom& g = hp.g;
  if (ndim < 2 || !nk_fast_size(g.nl) || !nk_fast_contig_ok(g.h)) return -100;
  auto tw_a = conv_tw<T>(hp.tw_a), twr = conv_tw<T>(hp.twr_a), tw_b = conv_tw<T>(hp.tw_b), tw_c = conv_tw<T>(hp.tw_c),
       tw_f = conv_tw<T>(hp.tw_f);
  const int sign = convention == NK_HARTLEY_CANONICAL ? -1 : 1;
  const NkPipe3 q = nk_pipe3_setup<T>(hp, sign, 2080, scale_first * (f->mul_scalar != 0.0 ? f->mul_scalar : 1.0));
  if (!nk_fast_strided_ok<T>(g.na, q.pm.s.inner) || (ndim == 3 && !nk_fast_strided_ok<T>(g.nm, q.rs))) return -100;
  if (nk_pipe3_work_elems(g, nk_pipe3_colpad<T>(), 2080) * sizeof(C2<T>) > hp.work_bytes) return -101;
  // poison the work array: every element a later pass reads must have been written by an earlier one
  std::vector<C2<T>> work(hp.work_bytes / sizeof(C2<T>) + 1, C2<T>{(T)NAN, (T)NAN});
  double energy = 0.0;
  switch (g.h) {
#define NK_CASE(NN) \
  case NN:          \
    emu3_contig<T, NN>(q.p1, *f, tw_a.data(), twr.data(), work.data()); \
    break;
    NK_FAST_SIZES(NK_CASE)
#undef NK_CASE
  }
  auto mid_axis = [&]() {
    switch (g.nm) {
#define NK_CASE(NN) \
  case NN:          \
    emu2_strided<T, NN>(q.s2, 0, *f, tw_b.data(), work.data(), nullptr, &energy); \
    break;
      NK_FAST_SIZES(NK_CASE)
#undef NK_CASE
    }
  };
  if (ndim == 3) mid_axis();
  switch (g.na) {
#define NK_CASE(NN) \
  case NN:          \
    emu3_mid<T, NN>(q.pm, *f, tw_c.data(), work.data(), cx); \
    break;
    NK_FAST_SIZES(NK_CASE)
#undef NK_CASE
  }
  if (ndim == 3) mid_axis();
  switch (g.nl) {
#define NK_CASE(NN) \
  case NN:          \
    emu3_final<T, NN>(q.pf, *f, tw_f.data(), work.data(), &energy); \
    break;
    NK_FAST_SIZES(NK_CASE)
#undef NK_CASE
  }
  if ((f->epi == NK_EPI_LIKELIHOOD || f->epi == NK_EPI_VJP) && f->value) *f->value += energy;
  return NK_OK;
}

extern "C" int emu4_hartley_sandwich(int ndim, const int64_t* shape, int dtype, int64_t batch, const nk_fuse* f,
                                     double scale_first, int convention, int cx) {
  if (dtype == NK_F32) return emu4_run<float>(ndim, shape, dtype, batch, f, scale_first, convention, cx);
  return emu4_run<double>(ndim, shape, dtype, batch, f, scale_first, convention, cx);
}

// strided-first pipeline (ndim >= 2)
template <typename T>
static int emu3_run(int ndim, const int64_t* shape, int dtype, int64_t batch, const nk_fuse* f, int convention) {
  NkHostPlan hp;
  const char* msg;
  int rc = nk_host_plan_init(hp, ndim, shape, dtype, batch, &msg);
  if (rc != NK_OK) return rc;
  if (ndim < 2) return -99;
  auto tw_b = conv_tw<T>(hp.tw_b), tw_c = conv_tw<T>(hp.tw_c), tw_f = conv_tw<T>(hp.tw_f);
  const NkGeom& g = hp.g;
  if (!nk_fast_size(g.nl)) return -100;
  if (ndim == 3 && !nk_fast_strided_ok<T>(g.nm, hp.pb.inner)) return -101;
  if (!nk_fast_strided_ok<T>(g.na, hp.pc.inner)) return -102;
  double energy = 0.0;
  std::vector<C2<T>> work(hp.work_bytes / sizeof(C2<T>) + 1);
  NkPipe2 pq = nk_pipe2_setup(hp, convention == NK_HARTLEY_CANONICAL ? -1 : 1, nk_env_int("NK_WORK_BLO", 0),
                              nk_env_int("NK_WORK_PAD", 2080));
  NkPassS pb = pq.s1, pc = ndim == 3 ? pq.s0 : pq.s1;
  if (ndim == 3) {
    switch (g.nm) {
#define NK_CASE(NN) \
  case NN:          \
    emu2_strided<T, NN>(pb, 3, *f, tw_b.data(), work.data(), nullptr, &energy); \
    break;
      NK_FAST_SIZES(NK_CASE)
#undef NK_CASE
    }
  }
  int tl_n1 = 0, tl_n2 = 0;
  if (nk_tl_split<T>(g, tl_n1, tl_n2)) {
    // two-level first-axis pass (nk_fft_t.hip: nk_tl_first_axis): n1-point sub-lines with the prologue, n2-point ones in place
    auto tw1 = conv_tw<T>(hp.tw_t64), tw2 = conv_tw<T>(tl_n2 == 64 ? hp.tw_t64 : hp.tw_t32);
    emu_tl<T, 64, 4>(pc, tl_n2, *f, tw1.data(), tw_c.data(), work.data());
    if (tl_n2 == 64) emu_tl<T, 64, 5>(pc, tl_n1, *f, tw2.data(), tw_c.data(), work.data());
    else emu_tl<T, 32, 5>(pc, tl_n1, *f, tw2.data(), tw_c.data(), work.data());
  } else
  switch (g.na) {
#define NK_CASE(NN) \
  case NN:          \
    emu2_strided<T, NN>(pc, ndim == 3 ? 0 : 3, *f, tw_c.data(), work.data(), nullptr, &energy); \
    break;
    NK_FAST_SIZES(NK_CASE)
#undef NK_CASE
  }
  const NkPassF pf = pq.pf;
  switch (g.nl) {
#define NK_CASE(NN) \
  case NN:          \
    emu2_final<T, NN>(pf, *f, tw_f.data(), work.data(), &energy); \
    break;
    NK_FAST_SIZES(NK_CASE)
#undef NK_CASE
  }
  if ((f->epi == NK_EPI_LIKELIHOOD || f->epi == NK_EPI_VJP) && f->value) *f->value += energy;
  return NK_OK;
}

extern "C" int emu3_hartley_fused(int ndim, const int64_t* shape, int dtype, int64_t batch, const nk_fuse* f,
                                  int convention) {
  if (dtype == NK_F32) return emu3_run<float>(ndim, shape, dtype, batch, f, convention);
  return emu3_run<double>(ndim, shape, dtype, batch, f, convention);
}

template <typename T>
static int emu2_run(int ndim, const int64_t* shape, int dtype, int64_t batch, const nk_fuse* f, int convention) {
  NkHostPlan hp;
  const char* msg;
  int rc = nk_host_plan_init(hp, ndim, shape, dtype, batch, &msg);
  if (rc != NK_OK) return rc;
  auto tw_a = conv_tw<T>(hp.tw_a), twr = conv_tw<T>(hp.twr_a), tw_b = conv_tw<T>(hp.tw_b), tw_c = conv_tw<T>(hp.tw_c);
  const NkGeom& g = hp.g;
  if (ndim != 1) return -99;  // >= 2-D fast transforms use the strided-first pipeline (emu3)
  if (!nk_fast_contig_ok(g.h)) return -100;
  NkPassA pa = hp.pa;
  pa.g.sign = convention == NK_HARTLEY_CANONICAL ? -1 : 1;
  double energy = 0.0;
  std::vector<C2<T>> work(hp.work_bytes / sizeof(C2<T>) + 1), scratch(hp.scratch_bytes / sizeof(C2<T>) + 1);
  switch (g.h) {
#define NK_CASE(NN) \
  case NN:          \
    emu2_contig<T, NN>(pa, ndim == 1, *f, tw_a.data(), twr.data(), work.data(), &energy); \
    break;
    NK_FAST_SIZES(NK_CASE)
#undef NK_CASE
  }
  if ((f->epi == NK_EPI_LIKELIHOOD || f->epi == NK_EPI_VJP) && f->value) *f->value += energy;
  return NK_OK;
}

extern "C" int emu2_hartley_fused(int ndim, const int64_t* shape, int dtype, int64_t batch, const nk_fuse* f,
                                  int convention) {
  if (dtype == NK_F32) return emu2_run<float>(ndim, shape, dtype, batch, f, convention);
  return emu2_run<double>(ndim, shape, dtype, batch, f, convention);
}

extern "C" int emu_hartley_fused(int ndim, const int64_t* shape, int dtype, int64_t batch, const nk_fuse* f,
                                 int convention) {
  if (dtype == NK_F32) return emu_run<float>(ndim, shape, dtype, batch, f, convention);
  return emu_run<double>(ndim, shape, dtype, batch, f, convention);
}

extern "C" int emu_plan_info(int ndim, const int64_t* shape, int dtype, int64_t batch, int64_t* info) {
  NkHostPlan hp;
  const char* msg;
  int rc = nk_host_plan_init(hp, ndim, shape, dtype, batch, &msg);
  if (rc != NK_OK) return rc;
  info[0] = hp.pa.tl.tile; info[1] = hp.threads_a; info[2] = (int64_t)hp.lds_a;
  info[3] = hp.pb.tl.tile; info[4] = hp.threads_b; info[5] = (int64_t)hp.lds_b;
  info[6] = hp.pc.tl.tile; info[7] = hp.threads_c; info[8] = (int64_t)hp.lds_c;
  info[9] = (int64_t)hp.work_bytes; info[10] = (int64_t)hp.scratch_bytes;
  return NK_OK;
}
