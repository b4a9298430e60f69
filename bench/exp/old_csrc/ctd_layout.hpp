// ctd_layout.hpp -- problem descriptor, LDS record layout and emit-code packing (host + device PODs).
//
// Restates the size/offset algebra of the reference's scheme structs:
//   Trapeze   src/ode/trapeze.jl:14-42      [X_1,U_1, .., X_N+1,U_N+1, V]                (:1-4)
//   Midpoint  src/ode/midpoint.jl:17-39     [X_1,U_1, .., X_N,U_N, X_N+1, V]             (:1-7)
//   Euler     src/ode/euler.jl:10-50        same layout as midpoint (explicit and implicit) (:1-8)
//   GL (cc)   src/ode/irk.jl:138-160        [X_i, U_i, K_i^1..K_i^s].., X_N+1, V         (:1-9)
//   GL (sw)   src/ode/irk_stagewise.jl:136-163  [X_i, U_i^1..U_i^s, K_i^1..K_i^s].., X_N+1, V  (:6-11)
// and the constraint layout [C_i^x, C_i^{k,1..s}, G_i].., G_N+1, B (irk_stagewise.jl:13-30).
// All offsets here are 0-based.
#pragma once
#include "ctd_common.hpp"

namespace ctd {

enum SchemeClass { SC_TRAPEZE = 0, SC_MIDPOINT = 1, SC_IRK = 2 };

constexpr int kMaxNV = 4;          // optimisation variables supported by the emit tables
constexpr int kMaxEdgeSlots = 6;   // step/node records the edge block can hold

struct Layout {
    int32_t scheme;        // CTD_SCHEME_*
    int32_t sc;            // SchemeClass
    int32_t s;             // stages (0 for trapeze / midpoint)
    int32_t stagewise;     // 1: one control per stage
    int32_t n, m, nv, p, bc;
    int32_t cu;            // control columns per step (m, or s*m stagewise)
    int32_t blk;           // _step_variables_block
    int32_t eqs;           // _state_stage_eqs_block
    int32_t cb;            // eqs + _step_pathcons_block
    int32_t final_control; // trapeze: U_{N+1} exists
    int32_t it0, itf;      // index of t0/tf in v, -1 = fixed
    int32_t free_time;     // it0 >= 0 || itf >= 0
    int64_t N;             // time steps
    int64_t nvar, ncon;
    int64_t v_off;         // nvar - nv : first optimisation variable
    double t0, tf;         // fixed values
    double a[9], b[3], c[3];   // Butcher tables (row-major a), Float64 arithmetic as in irk_stagewise.jl:61-64,103-109
    int32_t euler;         // SC_MIDPOINT class only: 0 midpoint, 1 explicit Euler, 2 implicit Euler (src/ode/euler.jl:10-50)
    int32_t cs;            // control_steps (DOCPtime, src/DOCP_data.jl:149): controls per time step, > 1 only for :midpoint
                           // (direct shooting, src/direct_shooting.jl:55-71; sub-step dynamics midpoint.jl:137-155)
};

// Butcher entries for a stage index j that differs between the lanes of a wave.  The tables sit in the kernel arguments:
// indexed with a vector register they cost a global load (and a full memory latency) per use, selected from the S scalar
// values they stay in scalar registers.  Same values, so the arithmetic is unchanged.
template <int S> CTD_HD double butcher_pick(const double* t, int j, int stride, int off) {
    const double t0 = t[off], t1 = t[(S > 1 ? stride : 0) + off], t2 = t[(S > 2 ? 2 * stride : 0) + off];   // uniform loads first
    if (S <= 1) return t0;
    if (S == 2) return j == 0 ? t0 : t1;
    return j == 0 ? t0 : (j == 1 ? t1 : t2);
}
template <int S> CTD_HD double butcher_a(const Layout& L, int j, int l) { return butcher_pick<S>(L.a, j, 3, l); }
template <int S> CTD_HD double butcher_b(const Layout& L, int j) { return butcher_pick<S>(L.b, j, 1, 0); }
template <int S> CTD_HD double butcher_c(const Layout& L, int j) { return butcher_pick<S>(L.c, j, 1, 0); }

// Workgroups whose ids are congruent modulo 8 share an XCD (and its L2) on MI355X.  With this bijective remap every XCD
// walks one contiguous run of tiles, so a tile's halo step and the cache line its output segment shares with the next tile
// stay inside one L2 (the id is a group label only; nothing depends on it for correctness).
CTD_HD int xcd_tile(int b, int nt) {
    const int q = nt / 8, r = nt % 8, k = b % 8;
    return (k < r ? k * (q + 1) : r * (q + 1) + (k - r) * q) + b / 8;
}

// doubles per record input of the edge block: own step block | X_{i+1} | U_{i+1} (trapeze) | U_{i-1} (implicit Euler's path control)
CTD_HD int edge_in_stride(const Layout& L) { return L.blk + L.n + 2 * L.m; }
// doubles between the staged step blocks of a tile: the block size rounded up to ODD.  The evaluating lanes read the same field
// of consecutive steps; with an even pitch they meet in a few LDS banks (12-state quadrotor, midpoint: 16 doubles = 128 bytes:
// every lane in one of two bank pairs, a 9-way conflict on each of the ~30 input reads of a lane)
CTD_HD int tile_in_stride(const Layout& L) { return L.blk | 1; }

// ---- per-step LDS record (doubles) -------------------------------------------------------------------
// [0] = 1.0
// C[kNC]     per-step coefficients the emit codes multiply with (same offset in every kind of record)
// S eval blocks (S = max(s,1)), each: F[n*n] G[n*m] W[n*nv] f[n] ft[n]
//     F = df/dx, G = df/du at the eval point, W = total d f/d v (explicit + through time and x_ij), f = value,
//     ft = df/dt (scratch)
// Sv[n*nv]   d(state-equation row)/dv
// Px[p*n] Pu[p*m] Pv[p*nv] Pt[p]   path-constraint Jacobian at the node (Pv total, Pt scratch)
// R[cb]      constraint values of the step (c rows)
// The final-path record (FP) uses the P* fields and R[0..p); the boundary record (B) uses its own fields below.  It is the
// LAST record of the edge block and has a size of its own (bsize): an OCP with many boundary rows (12-state quadrotor: 23 rows
// x 2 x 13 doubles) must not inflate the stride of every step record -- tiles of the one-point schemes held 8 steps where 17
// fit.
constexpr int kNC = 18;
enum { C_ZERO = 0, C_ONE = 1, C_NEG1 = 2, C_HA = 3 /* -h a_jl at 3+3j+l */, C_HB = 12 /* -h b_j */, C_B = 15 /* b_j */,
       C_NHH = 3 /* -h/2 (trapeze, midpoint) */, C_NH = 4 /* -h (midpoint) */ };

struct RecLayout {
    int32_t S, eval_sz;
    int32_t ldg;                          // row pitch of G: ldu, or (m control_steps) | 1 for the midpoint scheme with several controls per step
    int32_t ldx, ldu;                     // row pitch of the (rows x n) and (rows x m) derivative blocks: n, m rounded up to odd.
                                          // The CSC emission walks DOWN a column (fixed c, consecutive rows r): with an even
                                          // pitch the lanes of a wave hit a few LDS banks only (8-way conflicts for n = 8)
    int32_t oF, oG, oW, of, oft;          // inside an eval block
    int32_t oEval;                        // first eval block
    int32_t oSv, oPx, oPu, oPv, oPt, oR, oC;
    int32_t oB0, oBf, oBv, oBval;         // boundary record
    int32_t stride;                       // doubles per step / node / final-path record
    int32_t bsize;                        // doubles of the boundary record (>= stride)
    int32_t nF, nG;                       // SPARSE eval blocks (OCPs with generated dynamics code): slots of the F / G blocks = their
                                          // structural nonzeros (DynNZ, ctd_kernel_body.hpp); -1: dense n x ldx / n x ldg blocks
};

// gcols: columns of the control blocks G / Pu -- m, or m * control_steps for the midpoint scheme with several controls per step
// (one eval block per step then holds the SUM over the sub-steps of F, W, f, ft and the n x (m control_steps) block [G_1 .. G_cs])
// nF, nG >= 0: the F block holds nF slots and the G block nG slots per control block (the structural nonzeros of df/dx and df/du
// in row-major order); a 12-state quadrotor step record on Gauss-Legendre 3 shrinks from 857 to 307 doubles, so more than twice the
// steps fit a tile and the evaluating waves run with full lanes
constexpr RecLayout make_rec_layout(int n, int m, int nv, int p, int bc, int s, int cb, int gcols = -1, int nF = -1, int nG = -1) {
    RecLayout r{};
    r.S = s > 0 ? s : 1;
    r.ldx = n | 1; r.ldu = m | 1; r.ldg = (gcols > 0 ? gcols : m) | 1;
    r.nF = nF; r.nG = nG;
    const int gblocks = (m > 0 && gcols > 0) ? gcols / m : 1;          // control blocks side by side (control_steps)
    r.oF = 0; r.oG = nF >= 0 ? nF : n * r.ldx; r.oW = r.oG + (nG >= 0 ? nG * gblocks : n * r.ldg); r.of = r.oW + n * nv; r.oft = r.of + n;
    r.eval_sz = r.oft + n;
    r.oC = 1;
    r.oEval = 1 + kNC;
    r.oSv = r.oEval + r.S * r.eval_sz;
    r.oPx = r.oSv + n * nv;
    r.oPu = r.oPx + p * r.ldx;
    r.oPv = r.oPu + p * r.ldu;
    r.oPt = r.oPv + p * nv;
    r.oR = r.oPt + p;
    int end_step = r.oR + cb;
    r.oB0 = 1 + kNC; r.oBf = r.oB0 + bc * r.ldx; r.oBv = r.oBf + bc * r.ldx; r.oBval = r.oBv + bc * nv;
    int end_b = r.oBval + bc;
    r.stride = end_step;
    if ((r.stride & 1) == 0) r.stride += 1;     // odd stride: lanes that index consecutive records spread over LDS banks
    r.bsize = end_b > r.stride ? end_b : r.stride;
    return r;
}

// ---- 32-bit emit code ---------------------------------------------------------------------------------
//   value = rec[crec][oC + ci] * rec[drec][di] + beta
// bits  0-15 di, 16-21 ci, 22-23 beta (0: 0, 1: +1, 2: -1), 24-26 drec, 27-29 crec.
// Inside tile templates drec/crec are relative (0 = the segment's own step, 1 = the previous step); inside the edge
// list they are absolute record ids of the edge block.
CTD_HD uint32_t pack_code(int di, int ci, int beta, int drec, int crec) {
    return (uint32_t)di | ((uint32_t)ci << 16) | ((uint32_t)beta << 22) | ((uint32_t)drec << 24) | ((uint32_t)crec << 27);
}
CTD_HD int code_di(uint32_t c) { return (int)(c & 0xFFFFu); }
CTD_HD int code_ci(uint32_t c) { return (int)((c >> 16) & 0x3Fu); }
CTD_HD int code_beta(uint32_t c) { return (int)((c >> 22) & 0x3u); }
// inside tile templates a record code of kRecNext means the NEXT step's record (relative -1): implicit Euler's path rows of
// node i+1 sit in the columns of U_i
constexpr int kRecNext = 7;
CTD_HD int code_drec(uint32_t c) { const int r = (int)((c >> 24) & 0x7u); return r == kRecNext ? -1 : r; }
CTD_HD int code_drec_raw(uint32_t c) { return (int)((c >> 24) & 0x7u); }     // absolute record ids of the edge list
CTD_HD int code_crec(uint32_t c) { return (int)((c >> 27) & 0x7u); }

// exact x / d for the small operands of the emit loops (x * d < 2^32): q = umulhi(x, M), M = floor(2^32 / d) + 1 (d > 1)
struct FastDiv { uint32_t d, M; };
inline FastDiv make_fastdiv(uint32_t d) { return FastDiv{d, d > 1 ? (uint32_t)(((uint64_t)1 << 32) / d + 1) : 0u}; }
CTD_HD uint32_t fast_div(uint32_t x, FastDiv f) { return f.d > 1 ? (uint32_t)(((uint64_t)x * f.M) >> 32) : x; }

constexpr int64_t kEdgeCBit = (int64_t)1 << 62;   // edge_idx flag: the entry goes to c[], not vals[]

// ---- sharded iterate read in place (multi-GPU) ------------------------------------
// The time steps of one transcription are split over G shards (the loop being partitioned: src/DOCP_functions.jl:92-98);
// each shard keeps the variables of its own steps in a full-length buffer of its own device.  The few entries a shard
// reads from its neighbours -- the next shard's first node, the previous shard's last step block (one-point schemes),
// X_1 and X_{N+1} for the boundary rows -- are loaded by the kernels straight from the owner's buffer (peer-mapped over
// xGMI, or an IPC mapping of another process): no copy, no collective, no event in the evaluation.  The table lives in
// device memory; only the first / last tile of a shard and its edge block ever look at it.
constexpr int kMaxShards = 16;
struct XHalo {
    int32_t G, self;
    int64_t vbegin[kMaxShards + 1];    // shard k owns the variables [vbegin[k], vbegin[k+1]); vbegin[G] = v_off (v is replicated)
    const double* x[kMaxShards];       // full-length iterate buffer of shard k (entry `self` is not used: the kernel's own xu)
};
// Balanced split of N steps over G shards (the rule of ctd_create_sharded, ctd_shard_steps, dist.shard_steps): the first N % G
// shards hold one step more.  shard_of_step inverts it.
CTD_HD int64_t shard_begin(int64_t N, int G, int k) { const int64_t base = N / G, rem = N % G; return k * base + (k < rem ? k : rem); }
CTD_HD int shard_of_step(int64_t N, int G, int64_t step) {
    const int64_t base = N / G, rem = N % G, cut = rem * (base + 1);
    return step < cut ? (int)(step / (base + 1)) : (int)(rem + (step - cut) / base);
}
// Stitching the constraint vector (ctd_stitch_c): every rank sends its row block padded to `smax` = the longest block + the p + bc
// tail rows; position of global row r inside the gathered buffer [G][smax] (the tail rows come from the LAST rank: the only one
// that holds everything they read when the iterate is sharded)
CTD_HD int64_t stitch_src(int64_t r, int64_t N, int cb, int G, int64_t smax) {
    if (r < N * cb) {
        const int64_t step = r / cb;
        const int k = shard_of_step(N, G, step);
        return k * smax + (step - shard_begin(N, G, k)) * cb + (r - step * cb);
    }
    const int64_t last_rows = (N - shard_begin(N, G, G - 1)) * cb;
    return (int64_t)(G - 1) * smax + last_rows + (r - N * cb);
}

// buffer that holds variable g
CTD_HD const double* xsrc(const XHalo* hl, const double* xu, int64_t g) {
    const int G = hl->G;
    if (g >= hl->vbegin[G]) return xu;
    int k = 0;
    while (k + 1 < G && g >= hl->vbegin[k + 1]) ++k;
    return k == hl->self ? xu : hl->x[k];
}

// The same lookup for the constraint / Jacobian kernel from KERNEL ARGUMENTS: the only entries of other shards a shard ever reads are
// the previous shard's last step block, the next shard's first node, X_1 and X_{N+1}; with their four buffers (null = this shard's own)
// and the shard's own variable range in scalar registers, a boundary tile's loads of those entries are ONE memory round trip
// instead of two (table, then data) -- they sit on the critical path of a latency-bound kernel (two ranks on one GPU: 9.5 -> 8.9 us
// per step against 8.1 with the whole x in place)
struct XNear {
    int64_t own_lo, own_hi;     // variables of this shard: [own_lo, own_hi)
    int64_t prev_lo;            // own_lo - blk: the previous shard's last step block starts here
    int64_t last_lo, v_off;     // X_{N+1} starts at last_lo = N blk; v (replicated) at v_off
    const double *prev, *next, *first, *last;
};
CTD_HD const double* xnear(const XNear& nr, const double* xu, int64_t g) {
    if ((g >= nr.own_lo && g < nr.own_hi) || g >= nr.v_off) return xu;
    const double* p = g >= nr.own_hi ? (g >= nr.last_lo ? nr.last : nr.next) : (g >= nr.prev_lo ? nr.prev : nr.first);
    return p ? p : xu;
}
inline XNear make_xnear(const XHalo& t, int64_t blk, int64_t N, int64_t v_off) {
    XNear nr{};
    const int G = t.G, me = t.self;
    nr.own_lo = t.vbegin[me]; nr.own_hi = t.vbegin[me + 1];
    nr.prev_lo = nr.own_lo - blk; nr.last_lo = N * blk; nr.v_off = v_off;
    nr.prev = me > 0 ? t.x[me - 1] : nullptr;
    nr.next = me + 1 < G ? t.x[me + 1] : nullptr;
    nr.first = me > 0 ? t.x[0] : nullptr;
    nr.last = me + 1 < G ? t.x[G - 1] : nullptr;
    return nr;
}

// ---- kernel parameters (passed by value) ---------------------------------------------------------------
struct KParams {
    Layout L;
    RecLayout R;
    const double* tau;          // normalized grid on device (N+1), or nullptr: uniform, tau_i = i / N
    // tiling of the shard [step_begin, step_end)
    int32_t T;                  // steps per tile
    int32_t HL, HH;             // extra records a tile needs below / above its steps (midpoint: 1,0; trapeze: 0,1)
    int32_t ntiles;
    int32_t has_edge;           // block 0 is the edge block
    int32_t xcd_remap;          // tiles follow xcd_tile(block) instead of the block id
    int64_t step_begin, step_end;
    // regular CSC segments: step i in [reg_first, reg_last) owns vals[seg_base + (i - reg_first) * Lseg, +Lseg)
    const uint32_t* tmpl;
    int32_t Lseg;
    int32_t vr;                 // rows per step inside each V column
    FastDiv div_cb, div_Lseg, div_vr, div_blk;
    int64_t seg_base;
    int64_t reg_first, reg_last;
    // V columns: column k holds, for step i, vals[vcol_base[k] + i * vr, +vr) with codes vtmpl[k * vr ...]
    const uint32_t* vtmpl;
    int64_t vcol_base[kMaxNV];
    // edge entries (first step, last step, final block, tails): explicit (index, code) list
    const int64_t* edge_idx;
    const uint32_t* edge_code;
    int32_t edge_begin, edge_end;       // edge entries this shard emits: [edge_begin, edge_end) (irregular leading columns of its
    int32_t edge2_begin, edge2_end;     // own steps) and [edge2_begin, edge2_end) (tail rows of c; trailing columns: last shard)
    int32_t n_edge_slots;
    int32_t edge_fp, edge_b;            // record ids of the final-path and boundary records
    int32_t edge_slot_first, edge_slot_last;   // slots holding step 0 and step N-1
    int64_t edge_steps[kMaxEdgeSlots];
    // outputs (global indexing); either may be null
    double* c;
    double* vals;
    // diagnostics only (ctd_debug_stamps): when non-null, lane 0 of every workgroup stores 6 x {realtime, cycle}
    // stamps at the phase boundaries; null in every normal launch
    unsigned long long* stamps;
    // diagnostics only (env CTD_DEBUG_STOP): 0 = normal; k > 0: every workgroup returns after phase k (1 nothing, 2 load,
    // 3 eval, 4 fin) -- ablation timing, outputs are then incomplete
    int32_t debug_stop;
    int32_t stage_codes;        // the emit templates are copied to LDS once per workgroup (short periods that cost no occupancy)
    // EARLY EMISSION (Gauss-Legendre schemes, direct tiles): outputs that only read what the lead role writes -- the state rows of
    // c, their Jacobian entries (-1, -h b_j, +1: 36 of the 102 entries per step of Goddard / GL2) and d/dv -- are stored by the
    // lead wave WHILE the dynamics are still being evaluated; the emit phase behind the barrier then streams the rest.
    // pos = [late positions of the CSC period ..., early positions ...] (null: everything is emitted behind the barrier)
    const uint16_t* pos;
    int32_t n_late, n_early;    // positions of the period emitted behind the barrier / by the lead wave
    int32_t c_early, vr_early;  // leading rows of every step's block of c / of every step's slice of a V column that are early
    FastDiv div_late;
    // sharded iterate read in place: where the other shards' variables live (device table), or null: xu holds everything
    // this shard reads
    const XHalo* halo;
    XNear near;                 // (valid when halo is set: the same buffers, for the lookups of this kernel)
    // MULTI-TILE WORKGROUPS (staged driver, grids of several rounds): workgroup w walks the blocks w, w + wg_stride, ... of the
    // evaluation instead of one (0: one block per workgroup).  The emit templates, the optimisation variable and the lane's codes
    // are fetched once per workgroup, and the x slice of the NEXT tile is loaded into registers while the current one is emitted
    // (ctd_kernels.hpp: cons_jac_body).  The edge block (block 0) is never followed by a tile: it is the longest block.
    int32_t wg_stride;
    int32_t wt_store;       // 1: the emit phase's stores are write-through (sc1): small launches, see emit_store (ctd_kernel_body.hpp)
};

}  // namespace ctd
