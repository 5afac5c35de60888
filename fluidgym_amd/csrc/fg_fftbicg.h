// Host entry points of the fused row kernels of the Helmholtz-preconditioned BiCGStab (fg_fftbicg.hip), called by fg_bicgstab_solve.
#pragma once
#include "fg_internal.h"
#include "fg_bicg.h"

#if !FG_F64
bool fg_fbicg_ok(const fg_state* s);      // 2-D, periodic uniform x with the real-FFT basis, walls in y (and FG_BICG_PFUSED != 0)
int fg_fbicg_forward(fg_state* s, const BicgPtrs& q, int kind, int it, int fold, hipStream_t st);   // kind 0: FS(it) | 1: FP(it)
int fg_fbicg_inverse(fg_state* s, const BicgPtrs& q, int kind, int it, hipStream_t st);             // kind 0: IT(it) | 1: IV(it)
#endif
