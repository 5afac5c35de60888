// Multi-block, non-orthogonal PISO path: state shared by fg_mb_topo.hip (host-side
// topology / coefficient tables, built once per mesh) and fg_mb_step.hip (kernels + solvers).
//
// The reference resolves block connections, corner walks and metric interpolation inside every kernel for every cell
// on every call (PISO_multiblock_cuda_kernel.cu:329-492, 1926-2001, 2757-2874).  All of that depends on the mesh only,
// and one mesh serves the whole env batch, so here it is done ONCE on the host at fg_mb_finalize and the kernels read
// flat tables that are shared by all envs of the batch (they stay in L2 while the per-env fields stream from HBM):
//   nbr[f][i]          neighbour cell across face f (global index), or -1 - slot for a prescribed (FIXED) face
//   fcode[f][i]        which contravariant component of the neighbour continues this face's flux, and its sign
//   T[i]               Minv (row major) | det of the cell;  Tb[slot] the same for a boundary face
//   Vdiag, Voff        viscous part of the advection-diffusion matrix for nu = 1 (orthogonal + centre/direct cross terms)
//   KPp, KPn           pressure matrix entries as (coefficient of 1/A_P, coefficient of 1/A_N(f)) per slot and face
//   SV*, SP*           lagged corner terms of velocity / pressure as ELL operators on cells and boundary slots
#pragma once
#include <vector>

#include "fg_internal.h"

// mb_real: the scalar type of the multi-block translation units -- fields, tables, recurrence scalars, C ABI (the public header
// declares the same entry points with fg_real): float in libfluidgym_hip.so, double in the fp64 build (libfluidgym_hip_f64.so,
// -DFG_REAL_DOUBLE).  What is written for 32-bit words -- the four-cells-per-thread kernels (float4), the on-chip CG, the multilevel
// preconditioner: fg_f32 -- stays compiled and is switched off at run time in that build (fg_mb_create / fg_mb_finalize /
// fg_mb_set_multilevel).  (Until round 4 the fp64 build renamed the keyword `float` for these files.)
typedef float fg_f32;
#ifdef FG_REAL_DOUBLE
typedef double mb_real;
#define FG_MB_F64 1
#else
typedef float mb_real;
#define FG_MB_F64 0
#endif
// libm in the type of mb_real (the float names in the fp32 build: the device code stays what it was)
__host__ __device__ __forceinline__ mb_real mb_fabs(mb_real x) { return FG_MB_F64 ? (mb_real)fabs((double)x) : (mb_real)fabsf((float)x); }
__host__ __device__ __forceinline__ mb_real mb_fmax(mb_real a, mb_real b) { return FG_MB_F64 ? (mb_real)fmax((double)a, (double)b) : (mb_real)fmaxf((float)a, (float)b); }
__host__ __device__ __forceinline__ mb_real mb_fmin(mb_real a, mb_real b) { return FG_MB_F64 ? (mb_real)fmin((double)a, (double)b) : (mb_real)fminf((float)a, (float)b); }
__device__ __forceinline__ mb_real mb_rsqrt(mb_real x) { return FG_MB_F64 ? (mb_real)rsqrt((double)x) : (mb_real)rsqrtf((float)x); }
__host__ __device__ __forceinline__ mb_real mb_sqrt(mb_real x) { return FG_MB_F64 ? (mb_real)sqrt((double)x) : (mb_real)sqrtf((float)x); }

#define FG_CL_G 4   // workgroups per cluster of the cluster CG (fg_mb_cluster.hip)

#define FG_MB_FIXED 0
#define FG_MB_CONNECTED 1
#define FG_MB_PERIODIC 2

struct MbBound {
    int type = FG_MB_FIXED;
    int other = -1;
    int axes[3] = {0, 0, 0};
    int slot0 = -1;  // first boundary slot of a FIXED face (slots run over the face cells, lowest axis fastest)
};

struct MbBlock {
    int size[3] = {1, 1, 1};
    int offset = 0;
    int ncells = 0;
    std::vector<double> coords;  // [d][(nz+1)][ny+1][nx+1]
    MbBound bounds[6];
    std::vector<double> Minv;    // [ncells][d*d]
    std::vector<double> det;     // [ncells]
};

// device view handed to kernels by value
struct MbDev {
    int d, F, N, NB, B;
    int KC, KB, KPN;                // ELL widths: velocity corner terms over cells / boundary slots, pressure corner terms
    const int32_t* nbr;             // [F][N]
    const int32_t* fcode;           // [F][N]  (axis of the neighbour's component) | (negate << 2)
    const mb_real* T;                 // [N][d*d+1]
    const mb_real* Tb;                // [NB][d*d+1]
    const int32_t* bcell;           // [NB] owner cell
    const int32_t* bface;           // [NB] face of the owner cell
    const mb_real* Vdiag;             // [N]
    const mb_real* Voff;              // [F][N]
    const mb_real* KPp;               // [(F+1)][F][N]   slot 0 = diagonal, slot 1+g = face g
    const mb_real* KPn;               // [(F+1)][F][N]
    const int32_t* SVc_idx;         // [KC][N]
    const mb_real* SVc_w;             // [KC][N]
    const int32_t* SVb_idx;         // [KB][N]
    const mb_real* SVb_w;             // [KB][N]
    const int32_t* SP_idx;          // [KPN][N]
    const int32_t* SP_face;         // [KPN][N]
    const mb_real* SP_wp;             // [KPN][N]
    const mb_real* SP_wn;             // [KPN][N]
    const mb_real* yproj;             // [N] unit vector the CG residuals are kept orthogonal to (constant by default)
};

struct fg_mb_state {
    int d = 2, B = 1, N = 0, NB = 0, F = 4;
    mb_real nu = 0.f;
    int quirk_diag_offset = 1;  // computeConnectedPos(..., borderOffset = 1) on diagonal walks (K.cu:2152, 2658, 2825)
    int quirk_first_layer = 1;  // K.cu:1952
    int nonortho_flags = 25;    // CENTER_MATRIX | DIRECT_MATRIX | DIAGONAL_RHS (the simulation's mode) or 10 = DIRECT_RHS | DIAGONAL_RHS
    bool finalized = false;
    std::vector<MbBlock> blocks;
    // host copies of the tables (also exported for tests)
    std::vector<int32_t> h_nbr, h_fcode, h_bcell, h_bface;
    std::vector<mb_real> h_T, h_Tb;
    std::vector<mb_real> h_Vdiag, h_Voff, h_KPp, h_KPn, h_SVc_w, h_SVb_w, h_SP_wp, h_SP_wn;
    std::vector<int32_t> h_SVc_idx, h_SVb_idx, h_SP_idx, h_SP_face;
    bool host_only = false;  // created with device < 0: tables are built and readable, nothing touches a GPU (CPU tests)
    MbDev dev{};
    std::vector<void*> owned;  // device allocations
    // bound fields (caller-owned device memory)
    mb_real* velocity = nullptr;   // [B][d][N]
    mb_real* pressure = nullptr;   // [B][N]   pressure of the last solve (lagged corner terms read it)
    mb_real* bvel = nullptr;       // [B][d][NB]
    const mb_real* source = nullptr;  // [B][d][N] or null
    // work buffers
    mb_real *cc, *fb, *Cdiag, *Coff, *rA, *rhs, *ures, *hvec, *div, *Pdiag, *Poff, *pres, *Sdiag, *Soff;
    mb_real* w[8];   // Krylov work vectors: r, rw, p, v, t | s of the fused BiCGStab kernels | second p and v of their ping-pong pairs
    FgDacc* acc;   // [B d][MB_ACC] order-independent reduction accumulators (fg_internal.h)
    mb_real* sc;
    int32_t *flags, *best_it, *it_ctr;
    hipStream_t capture_stream = nullptr;
    hipGraphExec_t cg_graph_exec = nullptr;   // one chunk of CG iterations + convergence check (fg_mb_step.hip::mb_cg)
    unsigned char cg_graph_key_storage[256] = {0};
    int32_t* flags_pinned = nullptr;
    int32_t* sys_map_dev = nullptr;    // [B d] systems of a compacted launch (mb_bicgstab: MbSolve::sys_map); host copy in sys_map_pinned
    int32_t* sys_map_pinned = nullptr;
    int dbg_compact = 1;               // FG_MB_COMPACT=0: every launch over all systems
    FgPoll poll = {};             // host polls on pinned sequence words (fg_internal.h FgPoll; created with the pinned mirrors)
    int32_t* verified = nullptr;   // [B d] BiCGStab convergence verification (mb_bicgstab): 0 open, 1 true residual checked, 2 being checked
    fg_solve_info *info_dev, *info_pinned = nullptr;
    mb_real* yproj = nullptr;
    // on-chip CG (fg_mb_step.hip::k_mbc_onchip): neighbour table packed to 16 bits per face, (low half = even
    // face, high half = odd face, 0xFFFF = prescribed face); [N][F/2] words, built when N < 65535
    uint32_t* nbr16 = nullptr;
    // multilevel preconditioner of the on-chip CG (fg_mb_set_multilevel)
    uint16_t *ml_a4 = nullptr, *ml_parent4 = nullptr; mb_real *ml_d4g = nullptr, *ml_aci8 = nullptr; uint2 *ml_rect4 = nullptr, *ml_child8 = nullptr;
    int ml_n4 = 0, ml_n8 = 0; mb_real ml_geom_diag_sum = 0.f; bool ml_on = false;
    // right-preconditioned pressure BiCGStab (kernel form): a trial with exponential back-off (mb_pressure_bicgstab): after a failed
    // attempt the next ml_bicg_skip solves run plain, the back-off doubles with every failure (up to 256) and halves with every success
    int ml_bicg_attempts = 0, ml_bicg_failures = 0, ml_bicg_skip = 0, ml_bicg_backoff = 4;
    int dbg_ml_warmup = 0;   // FG_MB_ML_WARMUP: pressure solves a handle runs plain before its first multilevel attempt
    int ml_cap4 = 0, ml_cap8 = 0;   // capacity of the tables above (the on-chip CG takes at most 2048 / 512 aggregates, the kernel form 65535 / 2048)
    // work arrays of the kernel form (mb_ml_apply): aggregate sums [B][n4], coarse solution [B][n8], 1 / scale [B], M p and M s [B][N]
    mb_real *ml_r4 = nullptr, *ml_z8 = nullptr, *ml_scale = nullptr, *ml_mp = nullptr, *ml_ms = nullptr;
    uint16_t* ml_p8c = nullptr;   // the 8 x 8 aggregate of every CELL (parent4[a4[i]]): one table level less in the prolongation
    mb_real* ml_r4c = nullptr; uint32_t* ml_pos4 = nullptr;   // r4 once more, ordered by parent: [B][n8][4] (absent children stay 0), and the slot of every aggregate in it
    mb_real* Poff4 = nullptr;    // [B][N][4] pressure off-diagonals interleaved per cell (2-D), written by k_mb_pmatrix next to Poff
    // ILU(0) of the velocity matrix as the right preconditioner of the preconditioned rung (mb_ilu_*, fg_mb_step.hip): level
    // schedules of the two triangular solves from the neighbour table (host, once per mesh), factors and M p / M s per env
    int32_t *ilu_order_f = nullptr, *ilu_order_b = nullptr;   // [N] cells sorted by forward / backward level
    std::vector<int32_t> ilu_start_f, ilu_start_b;            // [levels + 1] first position of every level in the order
    int32_t *ilu_start_f_dev = nullptr, *ilu_start_b_dev = nullptr;
    mb_real *ilu_w = nullptr, *ilu_ud = nullptr;                 // [B][F][N] modified off-diagonals (l_ik below, u_ij above the diagonal), [B][N] u_ii
    mb_real *ilu_mp = nullptr, *ilu_ms = nullptr;                // [B][d][N]
    int ilu_state = 0;         // 0 not tried, 1 schedules built, -1 the mesh does not qualify (a cell with the same neighbour across two faces)
    int dbg_rung_ilu = 1;      // FG_MB_RUNG_ILU=0: the preconditioned rung keeps the right diagonal scaling of rounds 1-3
    // aggregate-owned layout of the on-chip CG (k_mbc_onchip<AGG>, built by fg_mb_set_multilevel when the mesh has at most 256
    // 8 x 8 aggregates of at most four 4 x 4 children of at most 16 cells): thread t = 4 * (8 x 8 aggregate) + child owns the
    // cells of that child, member m (row-major in its rectangle) lives in slot t + 1024 m of a 16 384-slot index space
    static constexpr int OC_SLOTS = 16 * 1024;
    int32_t* oc_slot_cell = nullptr;   // [OC_SLOTS] cell of a slot, -1 = hole
    uint16_t* oc_cell_slot = nullptr;  // [N] slot of a cell
    uint2* oc_nbr = nullptr;           // [OC_SLOTS] the four neighbour slots of a slot's cell as byte offsets (slot * 4), 16 bits each; prescribed face = the slot itself
    mb_real* oc_d4g = nullptr;           // [1024] 1 / diag(Z4^T S Z4) of the thread's aggregate (0: the thread owns none)
    int32_t* oc_cnt = nullptr;         // [1024] cells the thread owns
    int oc_rtg_nt = 1024;               // FG_MB_OC_RTG_NT: workgroup size of that instance (1024 x 24 cells or 512 x 48 cells per thread)
    mb_real* oc_rt_scratch = nullptr;   // [B][N] r - mean r of the on-chip preconditioner pass on meshes of 16-24 k cells (fg_mb_onchip.hip RTG)
    mb_real *Poff4s = nullptr, *Pdiag_s = nullptr, *oc_bestx = nullptr;   // [B][OC_SLOTS][4], [B][OC_SLOTS], [B][OC_SLOTS]: slot order; holes stay 0
    bool oc_agg = false;               // tables above installed
    bool oc_matrix_stale = true;       // no k_mb_pmatrix launch has written the slot-ordered matrix since the tables were installed
    int dbg_oc_agg = 1;                // FG_MB_OC_AGG=0: keep the cell-ordered on-chip kernel
    int oc_variant = 0;        // FG_MB_OC_VARIANT (tuning switches of the on-chip CG)
    unsigned long long* oc_dbg = nullptr;   // per-phase cycle counts (fg_mb_debug_cycles)
    int onchip_mode = 1;       // FG_MB_ONCHIP: 0 never, 1 when the mesh fits one workgroup's LDS / registers (default)
    // cluster CG (fg_mb_cluster.hip, round 6): FG_CL_G workgroups per env, each owning a contiguous range of the 8 x 8 aggregates;
    // slot s of workgroup g = thread + NT * member.  Tables built by mb_cluster_build behind fg_mb_set_multilevel
    int cl_mode = 1;           // FG_MB_CLUSTER: 0 never, 1 meshes beyond 8 k cells (default), 2 every mesh the tables fit
    int cl_near = 1;           // FG_MB_CL_NEAR=0: granule stores always write through (sc1), also when a cluster's workgroups share an XCD
    int cl_max_clusters = 0;   // FG_MB_CL_MAXCL: cap on the clusters of a launch (tests: envs beyond it queue inside the kernel)
    bool cl_on = false;        // tables installed
    bool cl_matrix_stale = true;   // no k_mb_pmatrix launch has written the cluster-ordered matrix since
    int cl_cpt = 0, cl_nt = 0, cl_S = 0, cl_W = 0, cl_n8g_max = 0, cl_n_out_max = 0, cl_n_halo_max = 0, cl_cus = 0;
    int cl_first[FG_CL_G + 1] = {0}, cl_n_out[FG_CL_G] = {0}, cl_n_halo[FG_CL_G] = {0};
    int32_t *cl_slot_cell = nullptr, *cl_out_slot = nullptr;
    uint16_t* cl_cell_slot = nullptr;     // [N] g * S + slot
    uint2* cl_nbr = nullptr;
    uint32_t *cl_tinfo = nullptr, *cl_halo_src = nullptr, *cl_epoch = nullptr, *cl_abort = nullptr;
    mb_real *cl_d4g = nullptr, *cl_cnt8 = nullptr, *cl_aci8 = nullptr, *cl_off4 = nullptr, *cl_diag = nullptr, *cl_bestx = nullptr;
    unsigned long long* cl_box = nullptr;
    uint16_t* cl_aci16 = nullptr; mb_real cl_aci16_unscale = 1.f;   // the coarse inverse as fp16 behind a power-of-two scale (meshes whose fp32 rows do not fit LDS)
    int cl_half = 1;           // FG_MB_CL_HALF=0: such meshes stream the fp32 rows from L2 instead
    int cl_WJ = 0; unsigned long long* cl_jbox = nullptr; uint32_t *cl_jepoch = nullptr, *cl_jabort = nullptr;   // granule boxes of the cluster sweeps (k_mbj_cluster)
    int cl_jacobi = 1;         // FG_MB_CL_JACOBI=0: the velocity sweeps stay one launch per sweep (k_mbj_sweep_env)
    long long cl_jacobi_solves = 0;
    long long cl_solves = 0, cl_fallbacks = 0;   // launches that solved / that a workgroup gave up on (repeated by the one-workgroup kernels)
    double* x64_best = nullptr; mb_real* best_res = nullptr; int32_t* best_keep = nullptr;   // its best refinement point
    double* x64 = nullptr;     // fp64 iterate of the refined BiCGStab (pressure_use_bicgstab = 2)
    // debug switches, read ONCE from the environment at fg_mb_create (never on the step path): FG_MB_BICG_VEC4 (per-kernel mask
    // of the four-cell BiCGStab kernels: 1 p, 2 v, 4 s, 8 t, 16 x; default all), FG_MB_SCALAR_CG=1 (one-cell
    // CG kernels), FG_MB_GRAPH (CG chunks replayed as a hipGraph), FG_MB_TRACE (residual trace on stderr)
    int dbg_vec_mask = 0, dbg_scalar_cg = 0, dbg_graph = 0, dbg_trace = 0, dbg_fail = 0;   // dbg_fail: FG_MB_TRACE_FAIL
    // iterations the last BiCGStab solve of the same place in the step took -- [0..3] velocity non-orthogonal pass,
    // [4 + 4 (c & 1) + (ps & 3)] pressure solve ps of corrector c (+ 16 for its multilevel-preconditioned attempt), [31] anything
    // else: where the next solve of that place polls first
    int pred_bicg[32] = {0};
    int dbg_ml_cap = 200;   // FG_MB_ML_TRY_CAP: iteration cap of a multilevel-preconditioned pressure BiCGStab attempt (tests: a tiny cap makes every attempt fail)
    int dbg_fuse_st = 2;   // FG_MB_BICG_FUSE: 0 five BiCGStab kernels, 1 s / t fused (k_mbb_st), 2 also p / v (k_mbb_pv; default)
    int dbg_ml_fuse = 1;   // FG_MB_ML_FUSE: multilevel BiCGStab forms p / s inside the restriction -- 0 never, 1 up to 32 systems (default), 2 always
    int dbg_ml_sb = 0;     // FG_MB_ML_SB: systems per workgroup of k_ml_coarse (4 / 8; 0 = 8 from 32 systems on)
    int dbg_pred = 1;      // FG_MB_PRED=0: first convergence poll after two iterations instead of where the previous solve finished
    // per-env outcome of the last fg_mb_piso_step / fg_mb_single_step: 0 ok, 1 a solve ended unconverged (best iterate used),
    // 2 a solve was non-finite: that env's step was NOT committed (state as before the step), the other envs completed
    mb_real* dt_step = nullptr;          // [B] working copy of the caller's dt; failed envs are masked out (dt = 0) in it
    int32_t* env_fail = nullptr;       // [B] device
    int32_t* env_fail_pinned = nullptr;
    std::vector<int32_t> env_status;   // [B] host, what fg_mb_env_status reports
    int cg_stall_limit = 400;  // fg_mb_set_stall_limit
    int adv_from_result = 0;   // fg_mb_set_advection_start
    // Jacobi sweeps for the velocity systems (mb_jacobi, fg_mb_krylov.hip): switch (fg_mb_set_advection_jacobi / FG_ADV_JACOBI), per
    // non-orthogonal pass the sweeps the last solve needed and the solves still to skip after a failure, pinned residuals of the check
    int adv_jacobi = 0, adv_jacobi_env = 0;
    int jac_sweeps[4] = {0, 0, 0, 0}, jac_skip[4] = {0, 0, 0, 0}, jac_fails[4] = {0, 0, 0, 0};
    mb_real* jac_res_pinned = nullptr;   // [B d][2]: residual at the last / the previous measuring sweep
    long long jac_solves = 0, jac_fallbacks = 0;
    bool yproj_const = true;   // yproj is the constant 1/sqrt(N): kernels use the scalar instead of loading it
    mb_real* red;        // [B] reductions (max)
    mb_real* pres_bak;   // [B][N] pressure at the start of a step (restored for envs whose step is dropped)
    mb_real* red8;       // [B][MB_SUM_WGS] per-workgroup partials of the pressure mean (summed in index order)
    mb_real* red_pinned = nullptr;
    mb_real *red2, *dt_dev;          // [2B] boundary flux sums, [B] time steps of the running substep
    mb_real *red2_pinned = nullptr, *dt_pinned = nullptr;
    int dt_slot = 0;              // half of dt_pinned the current sub-step's time steps were written to
    // live timing of the CG pair for bench.py's roofline (fg_mb_profile_*): the first iteration of sampled chunks is issued
    // with start/stop events on the kernels' own dispatch packets; active systems are known from the poll before the chunk
    int prof_on = 0, prof_used = 0, prof_chunk = 0;
    hipEvent_t prof_ev[2 * 32] = {nullptr};
    int prof_kind[32] = {0}, prof_active[32] = {0};
    // kind 0 / 1: the chunked CG's kernel pair; kind 2: the on-chip CG (one launch per solve; prof_its = its iterations)
    double prof_ms[3] = {0, 0, 0}, prof_bytes[3] = {0, 0, 0};
    long long prof_n[3] = {0, 0, 0}, prof_launches[3] = {0, 0, 0}, prof_its = 0;
    hipEvent_t prof_ev_oc[2] = {nullptr, nullptr};
    FgCounters ctr;    // iterations per solve kind since the last reset (fg_mb_solver_counters)
    long long ladder[4] = {0, 0, 0, 0};   // fg_mb_ladder: velocity fp64 rung, velocity preconditioned rung, pressure fp64 rung, pressure CG
    int ladder_force = 0;
    std::string err;
};

int fg_mb_build_tables(fg_mb_state* s);  // fg_mb_topo.hip
