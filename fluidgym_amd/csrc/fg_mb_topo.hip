// Host-side topology and coefficient tables of the multi-block path (see fg_mb.h).  Runs once per mesh.
//
// Restates, per cell, what the reference's device functions resolve on every call:
//   computeConnectedPos / computeConnectedDir / resolveNeighborCell   PISO_multiblock_cuda_kernel.cu:329-492  ("K.cu")
//   k_CoordsToTransforms, k_CoordsToFaceTransforms                     grid_gen.cu:298-354, 398-470
//   getLaplaceCoefficient*, interpolateNonOrthoLaplaceComponents       K.cu:1224-1506, 1926-2001
//   getCornerValue, getBlockDataNeighborDiagonal                       K.cu:2757-2874, 2630-2678
//   the coefficient logic of PISO_build_matrix / PISO_build_pressure_matrix / getNonOrthoLaplaceRHS_v2
//                                                                      K.cu:3616-3880, 4812-4978, 3048-3202
// with nonOrthoFlags = CENTER_MATRIX | DIRECT_MATRIX | DIAGONAL_RHS (PISOtorch_simulation.py:479-487), the only mode the
// simulation runs in.
#include <math.h>
#include <string.h>

#include <algorithm>
#include <array>
#include <map>

#include "fg_mb.h"

namespace {

struct Pos { int a[3]; };
using Mat = std::array<double, 9>;

struct Topo {
    fg_mb_state* s;
    int d, F;

    const MbBlock& blk(int b) const { return s->blocks[b]; }
    int flat(int b, const Pos& p) const {
        const int* z = blk(b).size;
        return p.a[0] + z[0] * (p.a[1] + z[1] * p.a[2]);
    }
    int gidx(int b, const Pos& p) const { return blk(b).offset + flat(b, p); }
    int face_flat(int b, int face, const Pos& p) const {
        const int axis = face >> 1;
        int idx = 0, stride = 1;
        for (int a = 0; a < d; ++a) {
            if (a == axis) continue;
            idx += stride * p.a[a];
            stride *= blk(b).size[a];
        }
        return idx;
    }
    bool at_bound(int b, const Pos& p, int face) const {
        const int a = face >> 1;
        return p.a[a] == ((face & 1) ? blk(b).size[a] - 1 : 0);
    }
    bool is_empty(int b, int face) const { return blk(b).bounds[face].type == FG_MB_FIXED; }
    int slot(int b, int face, const Pos& p) const { return blk(b).bounds[face].slot0 + face_flat(b, face, p); }

    Pos connected_pos(const Pos& p, int bdim, const MbBound& cb, int border_offset) const {
        const MbBlock& o = blk(cb.other);
        Pos c = p;
        int ca = cb.axes[0] >> 1;
        c.a[ca] = (cb.axes[0] & 1) ? o.size[ca] - 1 - border_offset : border_offset;
        for (int k = 1; k < d; ++k) {
            const int axis = (bdim + k) % d;
            ca = cb.axes[k] >> 1;
            c.a[ca] = (cb.axes[k] & 1) ? o.size[ca] - 1 - p.a[axis] : p.a[axis];
        }
        return c;
    }
    int connected_dir(int dir, int bdim, const MbBound& cb) const {
        const int rel = (((dir >> 1) - bdim) % d + d) % d;
        return cb.axes[rel] ^ (dir & 1);
    }

    struct Nb { int b; Pos p; int map[3]; };
    // resolveNeighborCell for a face that is not prescribed
    Nb neighbor(int b, const Pos& p, int face) const {
        Nb n;
        n.b = b; n.p = p;
        for (int a = 0; a < 3; ++a) n.map[a] = 2 * a;
        const int axis = face >> 1;
        if (at_bound(b, p, face)) {
            const MbBound& bd = blk(b).bounds[face];
            if (bd.type == FG_MB_CONNECTED) {
                n.b = bd.other;
                n.p = connected_pos(p, axis, bd, 0);
                for (int a = 0; a < d; ++a) n.map[a] = connected_dir(2 * a, axis, bd);
            } else {
                n.p.a[axis] = (face & 1) ? 0 : blk(b).size[axis] - 1;
            }
        } else {
            n.p.a[axis] += (face & 1) ? 1 : -1;
        }
        return n;
    }

    const double* Minv(int b, const Pos& p) const { return &blk(b).Minv[(size_t)flat(b, p) * d * d]; }
    double det(int b, const Pos& p) const { return blk(b).det[flat(b, p)]; }
    double alpha(int b, const Pos& p, int c1, int c2) const {
        const double* m = Minv(b, p);
        double v = 0;
        for (int k = 0; k < d; ++k) v += m[c1 * d + k] * m[c2 * d + k];
        return det(b, p) * v;
    }

    // ---- corner walk (getCornerValue with includeDepth0 = includeDepth1 = false, maxDepth = 2)
    struct Corner {
        int num;            // cells counted; 0: value comes from a Dirichlet boundary
        int ncell;          // depth-2 cells collected
        int cell[2];
        int n1;             // depth-1 cells (the two face neighbours next to the corner), for NON_ORTHO_DIRECT_RHS
        int cell1[2];
        int nslot;          // boundary slots (value = mean of them)
        int slotv[2];
    };
    Corner corner(int b, const Pos& p, int dir1, int dir2) const {
        Corner r{};
        r.num = 1;
        struct Cyc { int d1, d2, b; Pos p; } cyc[2] = {{dir1, dir2, b, p}, {dir2, dir1, b, p}};
        for (int depth = 1; depth <= 2; ++depth) {
            for (int k = 0; k < 2; ++k) {
                Cyc& c = cyc[k];
                const int axis = c.d1 >> 1;
                if (at_bound(c.b, c.p, c.d1)) {
                    const MbBound& bd = blk(c.b).bounds[c.d1];
                    if (bd.type == FG_MB_FIXED) {
                        r.num = 0; r.ncell = 0; r.n1 = 0;
                        r.nslot = 1;
                        r.slotv[0] = slot(c.b, c.d1, c.p);
                        if (!at_bound(c.b, c.p, c.d2)) {
                            Pos q = c.p;
                            q.a[c.d2 >> 1] += (c.d2 & 1) ? 1 : -1;
                            r.slotv[1] = slot(c.b, c.d1, q);
                            r.nslot = 2;
                        }
                        return r;
                    }
                    if (bd.type == FG_MB_CONNECTED) {
                        const Pos q = connected_pos(c.p, axis, bd, s->quirk_diag_offset);
                        const int d1 = c.d1;
                        c.d1 = connected_dir(c.d2, axis, bd);
                        c.d2 = connected_dir(d1, axis, bd) ^ 1;
                        c.b = bd.other;
                        c.p = q;
                    } else {
                        c.p.a[axis] = (c.d1 & 1) ? 0 : blk(c.b).size[axis] - 1;
                        const int d1 = c.d1;
                        c.d1 = c.d2;
                        c.d2 = d1 ^ 1;
                    }
                } else {
                    c.p.a[axis] += (c.d1 & 1) ? 1 : -1;
                    const int d1 = c.d1;
                    c.d1 = c.d2;
                    c.d2 = d1 ^ 1;
                }
                const Cyc& o = cyc[k ^ 1];
                if (c.b == o.b && c.p.a[0] == o.p.a[0] && c.p.a[1] == o.p.a[1] && c.p.a[2] == o.p.a[2]) return r;
                if (depth > 1) r.cell[r.ncell++] = gidx(c.b, c.p);
                else r.cell1[r.n1++] = gidx(c.b, c.p);
                ++r.num;
            }
        }
        return r;
    }
    // getBlockDataNeighbor (K.cu:2534-2578): -1 at a prescribed boundary; over a connection it lands borderOffset = 1 inside
    int neighbor_data(int b, const Pos& p, int dir) const {
        const int dim = dir >> 1;
        Pos q = p;
        if (at_bound(b, p, dir)) {
            const MbBound& bd = blk(b).bounds[dir];
            if (bd.type == FG_MB_FIXED) return -1;
            if (bd.type == FG_MB_CONNECTED) return gidx(bd.other, connected_pos(p, dim, bd, s->quirk_diag_offset));
            q.a[dim] = (dir & 1) ? 0 : blk(b).size[dim] - 1;
        } else {
            q.a[dim] += (dir & 1) ? 1 : -1;
        }
        return gidx(b, q);
    }
    // getBlockDataNeighborDiagonal: -1 if the walk ends on a prescribed boundary
    int neighbor_diagonal(int b, const Pos& p, int dir1, int dir2) const {
        const bool first_empty = is_empty(b, dir1);
        const int dirs[2] = {first_empty ? dir2 : dir1, first_empty ? dir1 : dir2};
        int cb = b;
        Pos cp = p;
        for (int i = 0; i < 2; ++i) {
            const int face = dirs[i], dim = face >> 1;
            if (at_bound(cb, cp, face)) {
                const MbBound& bd = blk(cb).bounds[face];
                if (bd.type == FG_MB_FIXED) return -1;
                if (bd.type == FG_MB_CONNECTED) {
                    cp = connected_pos(cp, dim, bd, s->quirk_diag_offset);
                    cb = bd.other;
                } else {
                    cp.a[dim] = (face & 1) ? 0 : blk(cb).size[dim] - 1;
                }
            } else {
                cp.a[dim] += (face & 1) ? 1 : -1;
            }
        }
        return gidx(cb, cp);
    }
    // cross metric halves on a face for the matrices (interpolateNonOrthoLaplaceComponents): {xP, xN}, both 0 if skipped
    bool cross_matrix(int b, const Pos& p, int face, int t, double& xP, double& xN) const {
        const int axis = face >> 1;
        xP = xN = 0;
        if (s->quirk_first_layer) {
            if (!((0 < p.a[axis] && p.a[axis] < blk(b).size[axis] - 1) || !is_empty(b, face))) return false;
        } else if (at_bound(b, p, face) && is_empty(b, face)) {
            return false;
        }
        xP = alpha(b, p, axis, t);
        const Nb n = neighbor(b, p, face);
        xN = alpha(n.b, n.p, t, axis);  // axes not mapped through a connection (K.cu:1957)
        return true;
    }
};

void invert(const double* M, int d, double* Minv, double& det) {
    if (d == 2) {
        det = M[0] * M[3] - M[1] * M[2];
        const double r = 1.0 / det;
        Minv[0] = M[3] * r; Minv[1] = -M[1] * r; Minv[2] = -M[2] * r; Minv[3] = M[0] * r;
    } else {
        const double a = M[0], b = M[1], c = M[2], dd = M[3], e = M[4], f = M[5], g = M[6], h = M[7], i = M[8];
        det = a * (e * i - f * h) - b * (dd * i - f * g) + c * (dd * h - e * g);
        const double r = 1.0 / det;
        Minv[0] = (e * i - f * h) * r; Minv[1] = (c * h - b * i) * r; Minv[2] = (b * f - c * e) * r;
        Minv[3] = (f * g - dd * i) * r; Minv[4] = (a * i - c * g) * r; Minv[5] = (c * dd - a * f) * r;
        Minv[6] = (dd * h - e * g) * r; Minv[7] = (b * g - a * h) * r; Minv[8] = (a * e - b * dd) * r;
    }
}

struct Coords {
    const MbBlock& b; int d;
    double at(int comp, int x, int y, int z) const {
        const int vx = b.size[0] + 1, vy = (d > 1 ? b.size[1] + 1 : 1), vz = (d > 2 ? b.size[2] + 1 : 1);
        return b.coords[(((size_t)comp * vz + z) * vy + y) * vx + x];
    }
};

void cell_transforms(MbBlock& b, int d) {
    const Coords C{b, d};
    b.Minv.assign((size_t)b.ncells * d * d, 0.0);
    b.det.assign(b.ncells, 0.0);
    for (int z = 0; z < b.size[2]; ++z)
        for (int y = 0; y < b.size[1]; ++y)
            for (int x = 0; x < b.size[0]; ++x) {
                double M[9] = {0};
                const int nv = 1 << d;
                for (int v = 0; v < nv; ++v) {
                    const int ox = v & 1, oy = (v >> 1) & 1, oz = (v >> 2) & 1;
                    for (int k = 0; k < d; ++k) {
                        const double sgn = ((v >> k) & 1) ? 1.0 : -1.0;
                        for (int i = 0; i < d; ++i) M[i * d + k] += sgn * C.at(i, x + ox, y + oy, z + oz);
                    }
                }
                const double norm = 1.0 / (double)(1 << (d - 1));
                for (int q = 0; q < d * d; ++q) M[q] *= norm;
                const int f = x + b.size[0] * (y + b.size[1] * z);
                invert(M, d, &b.Minv[(size_t)f * d * d], b.det[f]);
            }
}

// boundary-face transform of the face cell at position p (face axis ignored)
void face_transform(const MbBlock& b, int d, int face, const Pos& p, double* Minv, double& det) {
    const Coords C{b, d};
    const int axis = face >> 1, upper = face & 1;
    const int n = b.size[axis];
    const int p_lo = upper ? n - 1 : 0, p_hi = p_lo + 1, p_face = upper ? n : 0;
    int tang[2];
    for (int i = 1; i < d; ++i) tang[i - 1] = (axis + i) % d;
    const int nfv = 1 << (d - 1);
    double M[9] = {0};
    auto vert = [&](int comp, int plane, int bits) {
        int q[3] = {p.a[0], p.a[1], p.a[2]};
        q[axis] = plane;
        for (int j = 0; j < d - 1; ++j) q[tang[j]] += (bits >> j) & 1;
        return C.at(comp, q[0], d > 1 ? q[1] : 0, d > 2 ? q[2] : 0);
    };
    for (int bits = 0; bits < nfv; ++bits)
        for (int i = 0; i < d; ++i) M[i * d + axis] += (vert(i, p_hi, bits) - vert(i, p_lo, bits)) / nfv;
    const int nev = (d > 2) ? 2 : 1;
    for (int j = 0; j < d - 1; ++j)
        for (int bits = 0; bits < nfv; ++bits) {
            const double sgn = ((bits >> j) & 1) ? 1.0 : -1.0;
            for (int i = 0; i < d; ++i) M[i * d + tang[j]] += sgn * vert(i, p_face, bits) / nev;
        }
    invert(M, d, Minv, det);
}

template <typename T>
int upload(fg_mb_state* s, const std::vector<T>& h, const T** out) {
    void* p = nullptr;
    const size_t bytes = (h.empty() ? 1 : h.size()) * sizeof(T);
    FG_HIP_CHECK(hipMalloc(&p, bytes));
    s->owned.push_back(p);
    if (!h.empty()) FG_HIP_CHECK(hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    *out = (const T*)p;
    return FG_OK;
}

}  // namespace

int fg_mb_build_tables(fg_mb_state* s) {
    const int d = s->d, F = 2 * d;
    Topo tp{s, d, F};
    // ---- offsets, cell transforms, boundary slots
    int off = 0, nb = 0;
    for (MbBlock& b : s->blocks) {
        b.offset = off;
        off += b.ncells;
        cell_transforms(b, d);
        for (int k = 0; k < b.ncells; ++k)
            if (!(b.det[k] > 0.0)) {
                fg_set_error("fg_mb_finalize: a cell has a non-positive Jacobian (left-handed or degenerate block)");
                return FG_ERR_INVALID_ARG;
            }
        for (int f = 0; f < F; ++f)
            if (b.bounds[f].type == FG_MB_FIXED) {
                b.bounds[f].slot0 = nb;
                nb += b.ncells / b.size[f >> 1];
            }
    }
    const int N = s->N = off;
    const int NB = s->NB = nb;
    const int tw = d * d + 1;
    s->h_T.assign((size_t)N * tw, 0.f);
    s->h_Tb.assign((size_t)(NB ? NB : 1) * tw, 0.f);
    s->h_nbr.assign((size_t)F * N, -1);
    s->h_fcode.assign((size_t)F * N, 0);
    s->h_bcell.assign(NB ? NB : 1, 0);
    s->h_bface.assign(NB ? NB : 1, 0);
    std::vector<double> Tb_d((size_t)(NB ? NB : 1) * tw, 0.0);
    std::vector<mb_real>&Vdiag = s->h_Vdiag, &Voff = s->h_Voff, &KPp = s->h_KPp, &KPn = s->h_KPn;
    Vdiag.assign(N, 0.f); Voff.assign((size_t)F * N, 0.f);
    KPp.assign((size_t)(F + 1) * F * N, 0.f); KPn.assign((size_t)(F + 1) * F * N, 0.f);
    struct CellTerm { int idx; double w; };
    struct PTerm { int idx, face; double wp, wn; };
    std::vector<std::vector<CellTerm>> svc(N), svb(N);
    std::vector<std::vector<PTerm>> spn(N);

    auto for_cells = [&](auto&& fn) {
        for (int b = 0; b < (int)s->blocks.size(); ++b) {
            const MbBlock& B = s->blocks[b];
            for (int z = 0; z < B.size[2]; ++z)
                for (int y = 0; y < B.size[1]; ++y)
                    for (int x = 0; x < B.size[0]; ++x) fn(b, Pos{{x, y, z}});
        }
    };
    // pass 1: transforms, neighbours, boundary slots
    for_cells([&](int b, const Pos& p) {
        const int g = tp.gidx(b, p);
        const double* mi = tp.Minv(b, p);
        for (int q = 0; q < d * d; ++q) s->h_T[(size_t)g * tw + q] = (mb_real)mi[q];
        s->h_T[(size_t)g * tw + d * d] = (mb_real)tp.det(b, p);
        for (int f = 0; f < F; ++f) {
            const int axis = f >> 1;
            if (tp.at_bound(b, p, f) && tp.is_empty(b, f)) {
                const int k = tp.slot(b, f, p);
                s->h_nbr[(size_t)f * N + g] = -1 - k;
                s->h_bcell[k] = g;
                s->h_bface[k] = f;
                double det;
                face_transform(s->blocks[b], d, f, p, &Tb_d[(size_t)k * tw], det);
                Tb_d[(size_t)k * tw + d * d] = det;
                continue;
            }
            const Topo::Nb n = tp.neighbor(b, p, f);
            s->h_nbr[(size_t)f * N + g] = tp.gidx(n.b, n.p);
            int code = axis;
            if (tp.at_bound(b, p, f) && s->blocks[b].bounds[f].type == FG_MB_CONNECTED) {
                const MbBound& cb = s->blocks[b].bounds[f];
                code = cb.axes[0] >> 1;
                if ((cb.axes[0] & 1) == (f & 1)) code |= 4;  // same-side connection: the neighbour's flux points the other way
            }
            s->h_fcode[(size_t)f * N + g] = code;
        }
    });
    for (size_t q = 0; q < Tb_d.size(); ++q) s->h_Tb[q] = (mb_real)Tb_d[q];
    auto alpha_b = [&](int k, int c1, int c2) {
        const double* m = &Tb_d[(size_t)k * tw];
        double v = 0;
        for (int q = 0; q < d; ++q) v += m[c1 * d + q] * m[c2 * d + q];
        return m[d * d] * v;
    };
    // pass 2: coefficient tables
    for_cells([&](int b, const Pos& p) {
        const int g = tp.gidx(b, p);
        double vdiag = 0, voff[6] = {0};
        auto kp = [&](int slotg, int f) -> mb_real* { return &KPp[((size_t)slotg * F + f) * N + g]; };
        auto kn = [&](int slotg, int f) -> mb_real* { return &KPn[((size_t)slotg * F + f) * N + g]; };
        for (int f = 0; f < F; ++f) {
            const int dim = f >> 1;
            const double fs = (f & 1) ? 1.0 : -1.0;
            const double aP = tp.alpha(b, p, dim, dim);
            if (tp.at_bound(b, p, f) && tp.is_empty(b, f)) {
                vdiag += 2.0 * aP;  // no-slip Dirichlet wall (K.cu:3832)
                // lagged tangential gradient of the boundary values (K.cu:3085-3128)
                const int k0 = tp.slot(b, f, p);
                for (int k = 1; k < d; ++k) {
                    const int t = (dim + k) % d;
                    const int nt = s->blocks[b].size[t];
                    Pos lo = p, hi = p;
                    double dist = 0.5;
                    if (p.a[t] != 0) lo.a[t] -= 1;
                    if (p.a[t] != nt - 1) hi.a[t] += 1;
                    if (p.a[t] == 0 || p.a[t] == nt - 1) dist = 1.0;
                    const double w = -fs * alpha_b(k0, t, dim) * dist;
                    svb[g].push_back({tp.slot(b, f, hi), w});
                    svb[g].push_back({tp.slot(b, f, lo), -w});
                }
                continue;
            }
            const Topo::Nb n = tp.neighbor(b, p, f);
            int comp = dim;
            if (tp.at_bound(b, p, f) && s->blocks[b].bounds[f].type == FG_MB_CONNECTED)
                comp = s->blocks[b].bounds[f].axes[0] >> 1;
            const double aN = tp.alpha(n.b, n.p, comp, comp);
            // orthogonal diffusion (K.cu:3717-3747) and pressure Laplacian (K.cu:4849-4885)
            vdiag += 0.5 * (aP + aN);
            voff[f] -= 0.5 * (aP + aN);
            *kp(0, f) -= (mb_real)(0.5 * aP); *kn(0, f) -= (mb_real)(0.5 * aN);
            *kp(1 + f, f) += (mb_real)(0.5 * aP); *kn(1 + f, f) += (mb_real)(0.5 * aN);
            for (int i = 1; i < d; ++i) {
                const int t = (dim + i) % d;
                // ---- matrices: centre + direct neighbours (K.cu:3749-3805, 4887-4936)
                double xP, xN;
                const bool in_matrix = (s->nonortho_flags & (1 | 16)) != 0, direct_rhs = (s->nonortho_flags & 2) != 0;
                if (in_matrix && tp.cross_matrix(b, p, f, t, xP, xN) && (xP != 0.0 || xN != 0.0)) {
                    const double a = 0.5 * (xP + xN);
                    for (int tu = 0; tu < 2; ++tu) {
                        const int tf = 2 * t + tu;
                        const double tfs = tu ? 1.0 : -1.0;
                        const Topo::Corner c = tp.corner(b, p, f, tf);
                        const bool has_t = s->h_nbr[(size_t)tf * N + g] >= 0, has_to = s->h_nbr[(size_t)(tf ^ 1) * N + g] >= 0;
                        if (c.num < 1) {
                            // velocity: Dirichlet value on the right-hand side, nothing here; pressure: one-sided
                            const double q = fs * tfs * 0.25;
                            *kp(0, f) += (mb_real)(3 * q * 0.5 * xP); *kn(0, f) += (mb_real)(3 * q * 0.5 * xN);
                            *kp(1 + f, f) += (mb_real)(3 * q * 0.5 * xP); *kn(1 + f, f) += (mb_real)(3 * q * 0.5 * xN);
                            if (has_to) { *kp(1 + (tf ^ 1), f) -= (mb_real)(q * 0.5 * xP); *kn(1 + (tf ^ 1), f) -= (mb_real)(q * 0.5 * xN); }
                        } else {
                            const double q = fs * tfs / (double)c.num;
                            vdiag -= q * a;
                            voff[f] -= q * a;
                            if (has_t) voff[tf] -= q * a;
                            *kp(0, f) += (mb_real)(q * 0.5 * xP); *kn(0, f) += (mb_real)(q * 0.5 * xN);
                            *kp(1 + f, f) += (mb_real)(q * 0.5 * xP); *kn(1 + f, f) += (mb_real)(q * 0.5 * xN);
                            if (has_t) { *kp(1 + tf, f) += (mb_real)(q * 0.5 * xP); *kn(1 + tf, f) += (mb_real)(q * 0.5 * xN); }
                        }
                    }
                }
                // ---- lagged corner terms (K.cu:3130-3196): the neighbour's metric IS mapped through the connection
                const double rP = tp.alpha(b, p, dim, t);
                const double* mn = tp.Minv(n.b, n.p);
                double rN = 0;
                for (int q = 0; q < d; ++q) rN += mn[(n.map[t] >> 1) * d + q] * mn[(n.map[dim] >> 1) * d + q];
                rN *= tp.det(n.b, n.p);
                for (int tu = 0; tu < 2; ++tu) {
                    const int tf = 2 * t + tu;
                    const double tfs = tu ? 1.0 : -1.0;
                    const Topo::Corner c = tp.corner(b, p, f, tf);
                    const double fa = 0.5 * (rP + rN);
                    if (c.num == 0) {
                        // velocity: Dirichlet corner value; pressure: one-sided through the opposite diagonal cell
                        for (int q = 0; q < c.nslot; ++q) svb[g].push_back({c.slotv[q], -fs * fa * tfs / c.nslot});
                        const int dg = tp.neighbor_diagonal(b, p, f, tf ^ 1);
                        if (dg >= 0) spn[g].push_back({dg, f, -fs * 0.5 * rP * (-tfs) * 0.25, -fs * 0.5 * rN * (-tfs) * 0.25});
                        if (direct_rhs) {  // one-sided difference through the face neighbour and the opposite tangential one (K.cu:3176-3179)
                            const int nf = tp.neighbor_data(b, p, f), nt = tp.neighbor_data(b, p, tf ^ 1);
                            if (nf >= 0) spn[g].push_back({nf, f, -fs * 0.5 * rP * tfs * 0.75, -fs * 0.5 * rN * tfs * 0.75});
                            if (nt >= 0) spn[g].push_back({nt, f, -fs * 0.5 * rP * (-tfs) * 0.25, -fs * 0.5 * rN * (-tfs) * 0.25});
                        }
                    } else {
                        for (int q = 0; q < c.ncell + (direct_rhs ? c.n1 : 0); ++q) {
                            const int cq = q < c.ncell ? c.cell[q] : c.cell1[q - c.ncell];
                            svc[g].push_back({cq, -fs * fa * tfs / c.num});
                            spn[g].push_back({cq, f, -fs * 0.5 * rP * tfs / c.num, -fs * 0.5 * rN * tfs / c.num});
                        }
                    }
                }
            }
        }
        Vdiag[g] = (mb_real)vdiag;
        for (int f = 0; f < F; ++f) Voff[(size_t)f * N + g] = (s->h_nbr[(size_t)f * N + g] >= 0) ? (mb_real)voff[f] : 0.f;
    });
    // ---- merge duplicate targets, drop zeros, pad to ELL
    auto merge = [](std::vector<CellTerm>& v) {
        std::map<int, double> m;
        for (const CellTerm& t : v) m[t.idx] += t.w;
        v.clear();
        for (auto& kv : m) if (kv.second != 0.0) v.push_back({kv.first, kv.second});
    };
    int KC = 0, KB = 0, KPN = 0;
    for (int g = 0; g < N; ++g) {
        merge(svc[g]); merge(svb[g]);
        std::vector<PTerm> keep;
        for (const PTerm& t : spn[g]) if (t.wp != 0.0 || t.wn != 0.0) keep.push_back(t);
        spn[g].swap(keep);
        KC = std::max(KC, (int)svc[g].size());
        KB = std::max(KB, (int)svb[g].size());
        KPN = std::max(KPN, (int)spn[g].size());
    }
    std::vector<int32_t>&c_idx = s->h_SVc_idx, &b_idx = s->h_SVb_idx, &p_idx = s->h_SP_idx, &p_face = s->h_SP_face;
    std::vector<mb_real>&c_w = s->h_SVc_w, &b_w = s->h_SVb_w, &p_wp = s->h_SP_wp, &p_wn = s->h_SP_wn;
    c_idx.assign((size_t)KC * N, 0); b_idx.assign((size_t)KB * N, 0); p_idx.assign((size_t)KPN * N, 0); p_face.assign((size_t)KPN * N, 0);
    c_w.assign((size_t)KC * N, 0.f); b_w.assign((size_t)KB * N, 0.f); p_wp.assign((size_t)KPN * N, 0.f); p_wn.assign((size_t)KPN * N, 0.f);
    for (int g = 0; g < N; ++g) {
        for (size_t k = 0; k < svc[g].size(); ++k) { c_idx[k * N + g] = svc[g][k].idx; c_w[k * N + g] = (mb_real)svc[g][k].w; }
        for (size_t k = 0; k < svb[g].size(); ++k) { b_idx[k * N + g] = svb[g][k].idx; b_w[k * N + g] = (mb_real)svb[g][k].w; }
        for (size_t k = 0; k < spn[g].size(); ++k) {
            p_idx[k * N + g] = spn[g][k].idx; p_face[k * N + g] = spn[g][k].face;
            p_wp[k * N + g] = (mb_real)spn[g][k].wp; p_wn[k * N + g] = (mb_real)spn[g][k].wn;
        }
    }
    MbDev& D = s->dev;
    D.d = d; D.F = F; D.N = N; D.NB = NB; D.B = s->B; D.KC = KC; D.KB = KB; D.KPN = KPN;
    if (s->host_only) return FG_OK;
    if (int rc = upload(s, s->h_nbr, &D.nbr)) return rc;
    if (int rc = upload(s, s->h_fcode, &D.fcode)) return rc;
    if (int rc = upload(s, s->h_T, &D.T)) return rc;
    if (int rc = upload(s, s->h_Tb, &D.Tb)) return rc;
    if (int rc = upload(s, s->h_bcell, &D.bcell)) return rc;
    if (int rc = upload(s, s->h_bface, &D.bface)) return rc;
    if (int rc = upload(s, Vdiag, &D.Vdiag)) return rc;
    if (int rc = upload(s, Voff, &D.Voff)) return rc;
    if (int rc = upload(s, KPp, &D.KPp)) return rc;
    if (int rc = upload(s, KPn, &D.KPn)) return rc;
    if (int rc = upload(s, c_idx, &D.SVc_idx)) return rc;
    if (int rc = upload(s, c_w, &D.SVc_w)) return rc;
    if (int rc = upload(s, b_idx, &D.SVb_idx)) return rc;
    if (int rc = upload(s, b_w, &D.SVb_w)) return rc;
    if (int rc = upload(s, p_idx, &D.SP_idx)) return rc;
    if (int rc = upload(s, p_face, &D.SP_face)) return rc;
    if (int rc = upload(s, p_wp, &D.SP_wp)) return rc;
    if (int rc = upload(s, p_wn, &D.SP_wn)) return rc;
    return FG_OK;
}
