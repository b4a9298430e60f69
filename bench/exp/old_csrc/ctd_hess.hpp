// ctd_hess.hpp -- Hessian of the Lagrangian: record layout, term codes and kernel parameters (host + device PODs).
//
// Reference: ADNLPModels serves hess_structure!/hess_coord!(nlp, x, y, vals; obj_weight) from nested-dual AD over the
// closures f and c! (backend selection src/collocation.jl:121-125) on the pattern CTDirect supplies with
// DOCP_Hessian_pattern (src/ode/trapeze.jl:240-303, midpoint.jl:240-300, irk.jl:423-496, irk_stagewise.jl:565-638);
// the solver sees the lower triangle of that pattern in CSC order.
//
// The engine uses the separability of the Lagrangian  L = obj_weight f + y' c : every nonlinear piece is an OCP
// function (dynamics, Lagrange cost, path, boundary, Mayer) evaluated at one point  zeta = (t, x, u, v)  that depends
// on the NLP variables of ONE time step through an (almost) linear map.  Per evaluation point the kernel forms the
// scalar  Phi = weights' F(zeta)  and its dense Hessian HD along the md = n + m + nv directions
//     x_0..x_{n-1} | u_0..u_{m-1} | V_0..V_{nv-1}
// with second-order forward numbers (ctd::Dual2).  The V directions are the total derivatives through the free time
// grid (get_time_grid, src/DOCP_data.jl:437-458): dzeta/dV_k = (dt/dv_k, dh/dv_k * sum_l a_jl K_l, 0, e_k).
// Each output entry is then a short sum of terms  coef1 * coef2 * record[di]  (chain rule of the scheme: X_i -> 1,
// K_i^l -> h a_jl, midpoint X_i / X_{i+1} -> 1/2, stagewise path control U_i^l -> b_l).
#pragma once
#include "ctd_layout.hpp"

namespace ctd {

// ---- per-step coefficients --------------------------------------------------------------------------------
constexpr int kHC = 36;
enum { HC_ONE = 0, HC_HALF = 1,
       HC_HA = 2,     // h a_jl            at HC_HA + 3 j + l
       HC_A = 11,     // a_jl              at HC_A + 3 j + l
       HC_B = 20,     // b_l               at HC_B + l
       HC_NBH = 23 }; // -b_l dh/dv_k      at HC_NBH + 3 k + l   (k < kMaxNV)

// a coefficient as constant * step-dependent factor: F(HF_ONE) = 1, F(HF_H) = h, F(HF_DH + k) = dh/dv_k
enum { HF_ONE = 0, HF_H = 1, HF_DH = 2 };

constexpr int kMaxPairs = 64;          // distinct coefficient products C[c1] * C[c2] the term tables may use
constexpr int kMaxTerms = 7;           // terms per output entry (S stage points of the step and of the one before + path point / state-equation row)
// Midpoint scheme with more than 3 controls per step (run-time OCPs): an X x X entry would sum 2 control_steps + 1 terms.  The
// second derivatives along directions that are not controls (X, V) are the same kind of term at every control's point, so a
// pass between evaluation and emission adds the blocks of points 1 .. S-1 to the block of point 0 for those pairs
// (hess_phase_stage_sum) and the term tables read point 0 only: at most 3 terms per entry whatever control_steps is.
constexpr bool hess_sums_stages(int sc, int cs) { return sc == SC_MIDPOINT && 2 * cs + 1 > kMaxTerms; }
constexpr int kMaxHessEdgeSlots = 6;   // step/node records of the edge block (3-bit record ids: + final path + boundary)

// ---- per-slot LDS record (doubles) ---------------------------------------------------------------------
// S stage blocks (S = max(s, 1)), each:  HD[md (md+1)/2]  (packed upper triangle, hess_tri)   RK[nv*n]  (Gauss-Legendre with free times:
//     RK[k*n + a] = h HD[x_a][V_k] + dh/dv_k dPhi/dx_a, the d2/dK dV_k entry up to the factor a_jl)
// HP[md (md+1)/2]   path point of the step (Gauss-Legendre / midpoint with path constraints)
// YX[n]       multipliers of the state-equation rows (Gauss-Legendre with free times: d2/dK^l dV_k of -h b_l y'K^l)
// (the products C[c1] C[c2] of the chain-rule coefficients live in a separate LDS table of npairs doubles per slot, cx.cp)
// The boundary record holds HB[mdb (mdb+1)/2], mdb = 2n + nv (directions x0 | xf | V); the final-path record uses HP.
// position of the pair {a, b} in the packed upper triangle (row-major, row <= column) of an md x md symmetric block: half the
// LDS of the full square, so twice the steps per tile at the same LDS budget
CTD_HD constexpr int hess_tri(int md, int a, int b) {
    return a <= b ? a * (2 * md - a - 1) / 2 + b : b * (2 * md - b - 1) / 2 + a;
}
CTD_HD constexpr int hess_tri_size(int md) { return md * (md + 1) / 2; }

struct HessRecLayout {
    int32_t md, mdb, S;
    int32_t stage_sz, oStage, oRK;   // stage block j at oStage + j * stage_sz: HD at +0, RK at +oRK
    int32_t oHP, oYX, oZero;        // rec[oZero] = 0.0: target of padded (absent) terms
    int32_t stride;
};

constexpr HessRecLayout make_hess_layout(int n, int m, int nv, int p, int sc, int s, bool free_time) {
    HessRecLayout r{};
    r.md = n + m + nv;
    r.mdb = 2 * n + nv;
    r.S = s > 0 ? s : 1;
    const bool rk = (sc == SC_IRK) && free_time;
    r.oRK = hess_tri_size(r.md);
    r.stage_sz = r.oRK + (rk ? nv * n : 0);
    r.oStage = 0;
    r.oHP = r.oStage + r.S * r.stage_sz;
    r.oYX = r.oHP + ((p > 0 && sc != SC_TRAPEZE) ? hess_tri_size(r.md) : 0);
    int end_step = r.oYX + (rk ? n : 0);
    int end_b = hess_tri_size(r.mdb);
    int body = end_step > end_b ? end_step : end_b;
    r.oZero = body;
    r.stride = body + 1;
    if ((r.stride & 1) == 0) r.stride += 1;
    return r;
}

// ---- parameters of the symbolically differentiated stage functions of a run-time OCP (ctd_sym.hpp, ctd_jit.cpp) -----
// The evaluation point as a function of the differentiation variables d = (dx[n], du[m], dv[nv]), all zero at the point:
//   t = T0 + sum_k TD_k dv_k      h = H0 + sum_k HD_k dv_k      x_r = X0_r + dx_r + sum_k HD_k KAP_r dv_k   (KAP: IRK only)
//   u_b = U0_b + du_b             v_k = V0_k + dv_k
//   Phi = sum_r W_r f_r + CL h l  (Gauss-Legendre stage)        Phi = h (sum_r W_r f_r + CL l)  (midpoint / Euler point)
//   trapeze node (shared by the steps before and after it, lengths hm and h; W = multipliers of the step after, WP of the
//   step before, WG of the node's path rows):
//   hm = HM0 + sum_k HMD_k dv_k     Phi = -1/2 sum_r f_r (hm WP_r + h W_r) + CL (hm + h) l + sum_r WG_r g_r
struct SymPrm { int T0, H0, CL, TD, HD, X0, KAP, U0, V0, W, HM0, HMD, WP, WG, count; };
constexpr SymPrm sym_prm(int n, int m, int nv, int np = 0) {
    SymPrm p{};
    p.T0 = 0; p.H0 = 1; p.CL = 2; p.TD = 3; p.HD = 3 + nv; p.X0 = 3 + 2 * nv; p.KAP = p.X0 + n; p.U0 = p.KAP + n;
    p.V0 = p.U0 + m; p.W = p.V0 + nv; p.HM0 = p.W + n; p.HMD = p.HM0 + 1; p.WP = p.HMD + nv; p.WG = p.WP + n;
    p.count = p.WG + np;
    return p;
}

constexpr int kSymStepChunk = 32;      // outputs of a step's assembly between two flushes (ctd_hess_step.hpp)

// parameters of the symbolically differentiated path point  Phi = sum_r WG_r g_r(t, x, u, v)  (SymPathH, lane-per-step kernel):
// t = T0 + sum_k TD_k dv_k, x = X0 + dx, u = U0 + du, v = V0 + dv; outputs the packed md x md triangle (hess_tri)
struct SymPathPrm { int T0, TD, X0, U0, V0, WG, count; };
constexpr SymPathPrm sym_path_prm(int n, int m, int nv, int np) {
    SymPathPrm p{};
    p.T0 = 0; p.TD = 1; p.X0 = 1 + nv; p.U0 = p.X0 + n; p.V0 = p.U0 + m; p.WG = p.V0 + nv; p.count = p.WG + np;
    return p;
}

// ---- 32-bit term code:  value += CP[pair] * rec[di]  of record `slot` ------------------------------------------
// bits 0-15 di, 16-23 pair id, 24-26 slot.  Inside tile templates slot is relative (0 = the entry's own step,
// 1 = the previous step, 2 = the next step); inside the edge lists it is the absolute record id of the edge block.
CTD_HD uint32_t pack_term(int di, int pair, int slot) {
    return (uint32_t)di | ((uint32_t)pair << 16) | ((uint32_t)slot << 24);
}
CTD_HD int term_di(uint32_t c) { return (int)(c & 0xFFFFu); }
CTD_HD int term_pair(uint32_t c) { return (int)((c >> 16) & 0xFFu); }
CTD_HD int term_slot(uint32_t c) { return (int)((c >> 24) & 0x7u); }
// The tile templates carry their terms as the two LDS offsets (doubles, signed 16 bit) from the entry's own slot: low half
// into the coefficient products (pair - sd * npairs), high half into the records (di - sd * stride), sd = +1 / -1 for the
// record of the previous / next step.
CTD_HD uint32_t pack_tile_term(int a_off, int b_off) { return ((uint32_t)a_off & 0xFFFFu) | ((uint32_t)b_off << 16); }
CTD_HD int tile_term_a(uint32_t c) { return (int)(c << 16) >> 16; }      // (sign-extending shifts)
CTD_HD int tile_term_b(uint32_t c) { return (int)c >> 16; }

// ---- kernel parameters -------------------------------------------------------------------------------------
struct HParams {
    Layout L;
    HessRecLayout R;
    const double* tau;          // normalized grid on device (N+1) or nullptr (uniform)
    int32_t T, HL, HH;          // steps per tile, records a tile needs below its first step (midpoint class: 1) and above its
                                // last one (implicit Euler with path constraints: 1)
    int32_t ntiles;
    int32_t n_edge_blocks;          // workgroups 0 .. n_edge_blocks - 1 share the edge entries, the tiles follow
    int64_t step_begin, step_end;   // shard of the time grid this launch evaluates (tiles cover [step_begin, step_end))
    int32_t xcd_remap;              // tiles follow xcd_tile(block - 1) (ctd_layout.hpp)
    int32_t edge_begin, edge_end;   // edge entries this shard emits: irregular leading columns of its own steps ...
    int32_t edge2_begin, edge2_end; // ... and the trailing columns (owner of step N-1)
    int32_t edge_vv;                // 1: this shard adds the V x V terms of the final-path / boundary / last-node points
    // regular CSC segments of the lower triangle: step i in [reg_first, reg_last) owns
    // vals[seg_base + (i - reg_first) * Lseg, +Lseg); entry e of the segment sums terms [tptr[e], tptr[e+1])
    // The tiles walk nc entries: entry k sits at position cpos[k] of the segment and sums the terms [tptr[k], tptr[k+1]).
    // compact = 0: all Lseg entries (position k).  Else only the entries that have terms; the others (structural zeros of
    // the pattern) are zero-filled: compact = 1 (segments that are mostly zeros): the tile zero-fills its whole part of vals
    // beforehand; compact = 2: it stores zeros at the nz positions zpos[] of every step.
    int32_t Lseg, nc, compact, nz;
    const uint32_t* tptr;       // nc + 1 offsets
    const uint32_t* cpos;       // nc positions (compact only)
    const uint32_t* zpos;       // nz positions (compact = 2)
    const uint32_t* terms;      // pack_tile_term codes
    int32_t nterms;
    int64_t seg_base, reg_first, reg_last;
    // V x V block: entry e (nvv = nv (nv+1)/2 of them, at vals[vv_idx[e]]) is a sum over ALL evaluation points; tile
    // contributions per step: terms [vptr[e], vptr[e+1]) of vterms (slot 0 = the step's own record)
    int32_t nvv;
    const uint32_t* vptr;
    const uint32_t* vterms;
    int32_t nvterms;
    int64_t vv_idx[kMaxNV * (kMaxNV + 1) / 2];
    // edge entries: explicit index + term range; the edge block also sums its share of the V x V entries
    // (terms [evptr[e], evptr[e+1]) of eterms)
    int32_t n_edge;
    const int64_t* edge_idx;
    const uint32_t* eptr;       // n_edge + 1
    const uint32_t* evptr;      // nvv + 1
    const uint32_t* eterms;
    int32_t n_edge_slots;
    int32_t edge_fp, edge_b;    // record ids of the final-path and boundary records
    int64_t edge_steps[kMaxHessEdgeSlots];
    // coefficient pairs: CP[i] = pair_c[i] * F(pairs[i] & 0xFF) * F(pairs[i] >> 8), i < npairs (pair 0 is ONE * ONE)
    int32_t npairs;
    uint16_t pairs[kMaxPairs];
    const double* pair_c;
    // eval tasks: outer direction p (bits 0-4) and up to 4 inner directions q_i (bits 5+5i, 31 = none) per lane
    const uint32_t* tasks;      // stage-type points
    const uint32_t* ptasks;     // path points
    const uint32_t* btasks;     // boundary + Mayer point
    int32_t ntask, nptask, nbtask;
    int32_t slot_tasks;         // S * ntask + nptask
    FastDiv div_ntask, div_stage_tasks, div_nptask, div_nc, div_nz, div_npairs;      // (stage_tasks = S * ntask)
    // inputs / outputs
    double obj_weight;
    double* vals;
    double* partials;           // (ntiles + n_edge_blocks) * nvv: V x V partial sums per workgroup (workgroup 0: the edge's)
    // sharded iterate read in place (ctd_set_x_shards; the multipliers y stay replicated): variables outside [own_lo, own_hi) -- the
    // previous shard's last block / the next shard's first node of a boundary tile, X_1 / X_{N+1} of the edge blocks -- are fetched
    // from the owners' buffers; null: xu holds everything
    const XHalo* halo;
    XNear near;
    int64_t own_lo, own_hi;
    // 1: the value stores are write-through (sc1): small launches leave nothing dirty in the L2s for the kernel boundary
    // (emit_store in ctd_kernel_body.hpp; profiles/r03_experiments.md)
    int32_t wt_store;
    // diagnostics only (env CTD_HESS_STOP): 0 normal; 1 return after load, 2 after eval (ablation timing, outputs incomplete)
    int32_t debug_stop;
    // diagnostics only (ctd_hess_debug_stamps): lane 0 of every workgroup stores 5 x {100 MHz realtime, shader cycles}
    unsigned long long* stamps;
};

}  // namespace ctd
