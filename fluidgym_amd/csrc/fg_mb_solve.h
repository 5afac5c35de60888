// Shared by the translation units of the multi-block Krylov solvers (fg_mb_step.hip: assembly kernels, BiCGStab / chunked CG kernels,
// multilevel preconditioner, step driver and C ABI; fg_mb_onchip.hip: the whole-solve-on-chip CG): accumulator names, the solver's
// pointer bundle, verdict bookkeeping, and the host entry points that cross the two files.
#pragma once
#include "fg_mb.h"

constexpr int MB_ACC = 12;  // doubles per system
constexpr int A_RHO = 0, A_RV = 2, A_SS = 3, A_TS = 4, A_TT = 5, A_RR = 6, A_SV = 7, A_ST = 8;  // A_SV, A_ST: sum v, sum t (projection)
// A_RHOE + (it & 1): the rho the recurrence of iteration `it` actually uses -- rw.r, or r.r after a breakdown restart (k_mbb_p)
constexpr int A_RHOE = 10;

constexpr int C_RHO = 0, C_PAP = 3, C_SUM = 8;  // C_SUM ring 8..10: yp . r_k (residual projection, see mb_cg)

constexpr int OC_N4 = 2048, OC_N8 = 512;

__device__ __forceinline__ bool mb_active(const mb_real* dt, int b) { return dt == nullptr || dt[b] > 0.f; }

// ---------------------------------------------------------------------------------------------------------------
// Krylov solvers on the ELL matrix (diag [B][N], off [B][F][N], shared neighbour table).  System sys = b * nc + comp
// is blockIdx.y; the scalars of the recurrences live in device accumulators (fp64) exactly as in the single-block
// solvers (fg_bicgstab.hip, fg_poisson.hip), so the host only polls convergence.
// ---------------------------------------------------------------------------------------------------------------
struct MbSolve {
    const mb_real* diag; const mb_real* off; const mb_real* rhs;
    mb_real* x; mb_real* r; mb_real* rw; mb_real* p; mb_real* v; mb_real* t;
    FgDacc* acc; mb_real* sc; int32_t* flags; fg_solve_info* info;
    int nc; mb_real tol;
    // best-iterate tracking of the CG pressure solve (returnBestResult, cg_solver_kernel.cu:345-361): sc[2 sys] holds the
    // residual of the kept iterate, best_it the iteration it belongs to, best_x the iterate itself
    mb_real* best_x; int32_t* best_it; int stall_limit;
    // device-side iteration index of the graph-replayed CG: ctr[0] read by k_mbc_ap*, ctr[1] - 1 by k_mbc_update*
    int32_t* it_ctr; int max_iterations;
    int it_base;  // BiCGStab: iteration index of the last restart (kernels run on the index since then, reports add this)
    // stall acceptance (off when 0): a system whose kept iterate is within accept_factor * tol and has not improved for
    // accept_window iterations ends with that iterate and counts as converged
    mb_real accept_factor; int accept_window;
    // BiCGStab on the singular pressure system: 1 = iterate on Q P with Q = I - 1 1^T / N (all vectors mean-free), which removes
    // the null space the plain recurrence breaks down on
    int project;
    // right preconditioning (multilevel, mb_ml_apply): when set, v = A mp with mp = M p, t = A ms with ms = M s, and the iterate
    // advances along mp / ms; the recurrence itself (p, s, r and all dot products) is the one of A M
    const mb_real* mp; const mb_real* ms;
    // fused s / t kernel (k_mbb_st*): s lives in its own buffer (the neighbours' s is recomputed from r and v, which must still be
    // there), k_mbb_x reads it from here and takes over the convergence-on-s decision; null = the separate s and t kernels
    mb_real* sbuf;
    // fused p / v kernel (k_mbb_pv*): p and v of the previous iteration (the neighbours' new p is recomputed from them, so the new p
    // and v go to the other buffer of a pair); q.p / q.v are the current ones.  Null = the separate p and v kernels
    const mb_real* p_prev; const mb_real* v_prev;
    // compacted launches (mb_bicgstab): when few systems of a batch still iterate, the per-iteration kernels are launched over those
    // only -- grid.y = n_map and the system of a workgroup row is sys_map[blockIdx.y] (nullptr: the identity, grid.y = all systems)
    const int32_t* sys_map; int n_map;
};

// accumulator / scalar / flag words: only through acc_ld / acc_st, sc_ld / sc_st, flag_ld / flag_st (fg_internal.h)
__device__ __forceinline__ mb_real mb_rms(double rr, int n) { return (mb_real)sqrt(rr / (double)n); }
// ok_flag: what a finite verdict stores (1 done; 4 = converged on s, the x kernel still owes x += alpha p) -- ONE store of the
// flag, after the info words (other workgroups of the env read it in the same launch)
__device__ __forceinline__ void mb_mark(const MbSolve& q, int sys, mb_real crit, int it, int ok_flag = 1) {
    const bool finite = isfinite(crit);
    q.info[sys].final_residual = crit;
    q.info[sys].used_iterations = it;
    q.info[sys].converged = finite ? 1 : 0;
    q.info[sys].is_finite = finite ? 1 : 0;
    flag_st(q.flags + (sys), finite ? ok_flag : 2);
}

#define MB_CELL                                         \
    const int i = blockIdx.x * FG_BLOCK + threadIdx.x;  \
    const int b = blockIdx.y;                           \
    const int N = D.N;                                  \
    const bool valid = i < N;

__device__ __forceinline__ mb_real mb_block_sum(mb_real v, mb_real* lds) {
    v = fg_wave_sum(v);
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = v;
    __syncthreads();
    const mb_real r = lds[0] + lds[1] + lds[2] + lds[3];
    __syncthreads();
    return r;
}
// NV sums with ONE barrier pair (lds: NV * 4 floats): in the launch-bound Krylov kernels the reduction tail is a visible
// share of the run time, and four sums one after the other are eight barriers
template <int NV>
__device__ __forceinline__ void mb_block_sums(mb_real (&v)[NV], mb_real* lds) {
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = fg_wave_sum(v[k]);
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < NV; ++k) lds[k * 4 + (threadIdx.x >> 6)] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = lds[k * 4] + lds[k * 4 + 1] + lds[k * 4 + 2] + lds[k * 4 + 3];
    __syncthreads();
}

// The accumulator adds of a reduction tail with one LANE per value (threads 0 .. NV-1 side by side) instead of thread 0 doing them
// one after the other: the exact split of a value into the FgDacc words and its atomics are a dependent chain of ~100
// instructions, and in launches whose workgroups are all resident at once that per-workgroup tail is exposed (round 4: 11.7 of
// 36.9 us in the single-block k_bicgf_b, profiles/micro_bicg2d.py).
template <int NV>
__device__ __forceinline__ void mb_acc_tail(FgDacc* a, const int (&slot)[NV], const mb_real (&val)[NV], const bool (&on)[NV]) {
    if (threadIdx.x < NV) {
        int sl = slot[0]; mb_real v = val[0]; bool o = on[0];
#pragma unroll
        for (int k = 1; k < NV; ++k)
            if ((int)threadIdx.x == k) { sl = slot[k]; v = val[k]; o = on[k]; }
        if (o) acc_add(a + sl, (double)v);
    }
}

#define MB_DISPATCH(s, ...)                    \
    do {                                       \
        if ((s)->d == 2) { constexpr int DIMS = 2; __VA_ARGS__ } \
        else { constexpr int DIMS = 3; __VA_ARGS__ }            \
    } while (0)

#define MB_DISPATCH_PM(s, pm, ...)                                             \
    do {                                                                        \
        if ((pm) == 0) { constexpr int PM = 0; MB_DISPATCH(s, __VA_ARGS__); }   \
        else if ((pm) == 1) { constexpr int PM = 1; MB_DISPATCH(s, __VA_ARGS__); } \
        else { constexpr int PM = 2; MB_DISPATCH(s, __VA_ARGS__); }              \
    } while (0)

template <typename T>
int mb_alloc(fg_mb_state* s, T** p, size_t count) {
    void* q = nullptr;
    FG_HIP_CHECK(hipMalloc(&q, (count ? count : 1) * sizeof(T)));
    FG_HIP_CHECK(hipMemset(q, 0, (count ? count : 1) * sizeof(T)));
    s->owned.push_back(q);
    *p = (T*)q;
    return FG_OK;
}

// host side (fg_mb_krylov.hip unless noted)
MbSolve mb_solve_ptrs(fg_mb_state* s, const mb_real* diag, const mb_real* off, const mb_real* rhs, mb_real* x, int nc, mb_real tol);
int mb_finish(fg_mb_state* s, int nsys, fg_solve_info* info_host, int* max_it);
// fg_mb_onchip.hip: the whole CG solve of every env in one launch (k_mbc_onchip); same arguments and results as mb_cg
bool mb_onchip_ok(const fg_mb_state* s, int pm_mode);
int mb_cg_onchip(fg_mb_state* s, const mb_real* dt, const mb_real* diag, const mb_real* off, const mb_real* rhs, mb_real* x, mb_real tol,
                 int max_iterations, int use_x0, int pm_mode, mb_real stall_accept, int* max_it, hipStream_t st);
// fg_mb_cluster.hip: the same solve by a cluster of workgroups per env (k_mbc_cluster); tables built behind fg_mb_set_multilevel
int mb_cluster_build(fg_mb_state* s, int n4, int n8, const int32_t* rect4_host, const uint16_t* parent4, const uint2* child8, const mb_real* rd4,
                     const mb_real* aci8_padded);
bool mb_cluster_ok(const fg_mb_state* s, int pm_mode, const mb_real* diag, const mb_real* off);
int mb_cg_cluster(fg_mb_state* s, const mb_real* dt, const mb_real* rhs, mb_real* x, mb_real tol, int max_iterations, int use_x0, int pm_mode,
                  mb_real stall_accept, int* max_it, hipStream_t st, bool* fell_back);
bool mb_jacobi_cluster_ok(const fg_mb_state* s, int nc);
int mb_jacobi_cluster(fg_mb_state* s, const mb_real* dt, const mb_real* diag, const mb_real* off, const mb_real* rhs, mb_real* x, mb_real tol,
                      int use_x0, hipStream_t st, bool* fell_back, bool* done);
constexpr int ML_N8_MAX = 2048, ML_ROWS = 16, ML_CG = 64;   // multilevel coarse solve: rows per workgroup, column groups
void mb_ml_scale(fg_mb_state* s, const mb_real* diag, hipStream_t st);
bool mb_ilu_prepare(fg_mb_state* s);
void mb_ilu_factor(fg_mb_state* s, const mb_real* dt, const mb_real* diag, const mb_real* off, hipStream_t st);
void mb_ilu_apply(fg_mb_state* s, const MbSolve& q, const mb_real* in, mb_real* out, hipStream_t st);
void mb_ml_apply(fg_mb_state* s, const MbSolve& q, const mb_real* in, mb_real* out, hipStream_t st, int fused = 0, int it = 0);
int mb_bicgstab(fg_mb_state* s, const mb_real* dt, const mb_real* diag, const mb_real* off, const mb_real* rhs, mb_real* x, int nc,
                mb_real tol, int max_iterations, int use_x0, int* max_it, hipStream_t st, int project = 0, int refine = 0, int multilevel = 0,
                int pred_slot = 31);
// point-Jacobi sweeps for the velocity systems where their rows are diagonally dominant (the cylinder meshes: 12 sweeps); *outcome: 0 not
// tried (backing off), 1 solved, 2 / 3 given up -- then the caller runs BiCGStab, from a cleared start vector (2) or from the sweeps' last
// iterate, which sits in x (3)
int mb_jacobi(fg_mb_state* s, const mb_real* dt, const mb_real* diag, const mb_real* off, const mb_real* rhs, mb_real* x, int nc, mb_real tol,
              int use_x0, int* max_it, hipStream_t st, int pred_slot, int* outcome);
int mb_pressure_bicgstab(fg_mb_state* s, const mb_real* dt, mb_real tol, int max_iterations, int use_x0, int* max_it, hipStream_t st, int project,
                         int refine, int pred_slot);
int mb_cg(fg_mb_state* s, const mb_real* dt, const mb_real* diag, const mb_real* off, const mb_real* rhs, mb_real* x, mb_real tol,
          int max_iterations, int use_x0, int project_mean, mb_real stall_accept, int* max_it, hipStream_t st);
