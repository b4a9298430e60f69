// ctd_host.hpp -- host-side model of a discretised OCP: sizes, grids, bounds, initial guess, sparsity pattern and
// the emit tables the kernels consume.  Pure C++ (no HIP): everything here is build-time work in the reference too
// (get_docp, src/collocation.jl:57-73).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "ctd_layout.hpp"
#include "ctd_hess.hpp"
#include "ctd_problems.hpp"

namespace ctd {

struct HostDesc {
    int problem, scheme, pattern_mode;
    int64_t grid_size;
    const double* time_grid;
    int64_t time_grid_len;
    int control_steps = 1;            // DOCP(ocp, grid_size, control_steps, scheme, time_grid), src/DOCP_data.jl:293
    int value_order = 0;              // ctd_desc.value_order: 0 = CSC (the reference's SparseArrays.sparse order), 1 = CSR
};

struct Block { int64_t r0, r1, c0, c1; };   // rows [r0,r1) x cols [c0,c1), 0-based

// Hessian of the Lagrangian: lower triangle of DOCP_Hessian_pattern in CSC order + the term tables of the Hessian kernel
// (ctd_hess_host.cpp)
struct HessModel {
    HessRecLayout R;
    int hk = 4;                       // inner directions per eval lane (ctd::HessK<P>)
    bool sym_stage = false;           // stage points use the OCP's symbolic second derivatives (run-time OCPs, ctd_sym.hpp)
    int64_t nnzh = 0;
    std::vector<Block> tail;          // blocks that do not belong to one step (already symmetrised)
    // regular part
    int Lseg = 0, HL = 0, HH = 0;
    int64_t seg_base = 0, reg_first = 0, reg_last = 0;
    std::vector<uint32_t> tptr, terms;            // Lseg + 1 offsets, term codes
    std::vector<int64_t> relrow;                  // row of every segment entry relative to its step block (V rows: 2^40 + k)
    std::vector<uint32_t> tcode;                  // the same terms as the tiles read them (pack_tile_term: LDS offsets)
    // what the tiles walk (HParams::compact): all entries, or -- segments that are mostly structural zeros of the pattern --
    // only those that have terms (cpos: their positions, ctptr: offsets into tcode)
    int compact = 0;
    std::vector<uint32_t> cpos, ctptr, zpos;
    std::vector<uint32_t> vptr, vterms;           // V x V contributions of one step
    int nvv = 0;
    int64_t vv_idx[kMaxNV * (kMaxNV + 1) / 2] = {0};
    // edge part
    std::vector<int64_t> edge_idx;
    std::vector<uint32_t> eptr, evptr, eterms;
    int edge_split = 0;               // edge entries [0, edge_split): head (leading irregular steps), the rest: tail
    std::vector<int> head_ptr;        // [reg_first + 1]: first head entry of every leading irregular step
    int n_edge_slots = 0, edge_fp = 0, edge_b = 0;
    int64_t edge_steps[kMaxHessEdgeSlots] = {0};
    // structural nonzeros of the evaluation points' dense Hessians (row-major md x md / mdb x mdb, upper triangle used)
    // and of the K x V helper RK (nv x n): found by pushing dependency masks through the OCP functions
    std::vector<uint8_t> need_stage, need_path, need_bnd, need_rk;
    // eval tasks (stage-type points, path points, boundary point)
    std::vector<uint32_t> tasks, ptasks, btasks;   // p | q_0 << 5 | q_1 << 10 | .. (31 = no direction)
    // coefficient pairs referenced by the term codes (pair 0 = ONE * ONE)
    std::vector<uint16_t> pairs;
    std::vector<uint16_t> pair_kind;  // per pair: step-dependent factors k1 | k2 << 8 (HF_*) ...
    std::vector<double> pair_c;       // ... and the constant factor: C[c1] C[c2] = pair_c * F(k1) * F(k2)
    // column starts (same scheme as the Jacobian's)
    std::vector<int64_t> cp_head, cp_tmpl, cp_tail;
};

struct Model {
    int problem = 0, pattern_mode = 0;
    // Order of the Jacobian VALUE array (ctd_desc.value_order).  0: CSC, the order of SparseArrays.sparse(Is, Js, ...) the reference
    // hands to ADNLPModels (midpoint.jl:229-232, irk_stagewise.jl:555-558).  1: CSR (north_star: "assembled ... in CSR on device"):
    // the rows of step i -- with their entries in the V columns inline -- are ONE contiguous range of Lseg values, so every field
    // below that speaks of "columns" then speaks of rows: tmpl / Lseg / seg_base / cp_* describe the row-periodic segment, vr = 0
    // (no separate V streams), and a shard of the grid owns one contiguous range of the value array
    int order = 0;
    ProblemInfo info;
    Layout L;
    RecLayout R;
    bool dyn_t = false, dyn_v = false;
    // lanes one step needs in the eval phase: dynamics passes per stage, path passes, fused lead lane (ctd_kernel_body.hpp Dirs<P>)
    int nch_dyn = 1, nch_path = 0;
    // sparse eval blocks (DynNZ, ctd_kernel_body.hpp): slot of d f_r / d x_c, d f_r / d u_c in the F / G block, -1 = structurally zero;
    // n_f < 0: dense blocks
    int n_f = -1, n_g = -1;
    std::vector<int> map_f, map_g;
    bool fused = true;
    // DOCPtime (src/DOCP_data.jl:147-152)
    bool uniform = true;
    std::vector<double> tau;
    // fixed grid (DOCP_data.jl:201-211): only meaningful when no time is free (zeros otherwise)
    double fixed_time(int64_t i) const { return L.free_time ? 0.0 : L.t0 + (tau[i] * (L.tf - L.t0)); }
    // DOCPbounds (src/DOCP_data.jl:235-240)
    void fill_bounds(double* var_l, double* var_u, double* con_l, double* con_u) const;     // ctd_bounds
    // pattern
    std::vector<Block> tail;
    int64_t nnzj = 0;
    int64_t dropped = 0;
    // regular (step-periodic) part
    std::vector<uint32_t> tmpl, vtmpl;
    int Lseg = 0, vr = 0;
    int64_t seg_base = 0, reg_first = 0, reg_last = 0;
    int64_t vcol_base[kMaxNV] = {0, 0, 0, 0};
    int HL = 0, HH = 0;
    // early emission (KParams::pos): positions of the period that only read the lead role's fields, late ones first
    std::vector<uint16_t> pos_order;
    int n_late = 0, n_early = 0, c_early = 0, vr_early = 0;
    // edge part: entries [0, edge_split) belong to the shards that own the leading irregular steps (step 0; every step when
    // N < 5: head_ptr), [edge_split, edge_split2) are the tail rows
    // of c (final-time path + boundary values: every shard computes them, x is replicated), the rest belongs to the owner
    // of step N-1
    std::vector<int64_t> edge_idx;
    std::vector<uint32_t> edge_code;
    int edge_split = 0, edge_split2 = 0;
    std::vector<int> head_ptr;        // [reg_first + 1]: first head entry of every leading irregular step (shard ownership)
    int n_edge_slots = 0, edge_fp = 0, edge_b = 0, edge_slot_first = 0, edge_slot_last = 0;
    int64_t edge_steps[kMaxEdgeSlots] = {0};
    // CSC column starts without materialising the pattern: explicit for the head columns [0, reg_first*blk) and the
    // tail columns [reg_last*blk, nvar); periodic in between (cp_tmpl is relative to the step's segment)
    std::vector<int64_t> cp_head, cp_tmpl, cp_tail;

    // CTD_PATTERN_OPTIMIZED: operator-level dependence masks of the OCP functions (bits: x 0..n-1, u n..n+m-1, v n+m..;
    // a dependence on t shows as the free-time variables).  Boundary masks: x0 0..n-1, xf n..2n-1, v 2n...
    std::vector<uint32_t> dep_f, dep_g, dep_b;
    bool opt_dep(int64_t row, int64_t col) const;     // (row, col) belongs to the optimized pattern

    struct Entry { int kind; int64_t cstep, dstep; int ci, di, beta; bool cconst; };   // kind 0 step row, 1 final path, 2 boundary
    Entry classify(int64_t row, int64_t col) const;
    void step_blocks(int64_t i, std::vector<Block>& out) const;
    void gen_column(int64_t j, std::vector<int64_t>& rows) const;
    void rows_from_blocks(int64_t j, const std::vector<Block>& cand, std::vector<int64_t>& rows) const;
    void gen_vcolumn_piece(int k, int64_t i0, int64_t i1, bool with_tail, std::vector<int64_t>& rows) const;
    int64_t column_start(int64_t j) const;    // CSC colptr[j] without materialising the pattern (order == 0)
    void gen_row(int64_t r, std::vector<int64_t>& cols) const;      // sorted columns of row r of the same pattern
    int64_t row_start(int64_t r) const;       // CSR rowptr[r] without materialising the pattern (order == 1)
    int64_t shard_vals_begin(int64_t step_begin) const;             // value range a shard [step_begin, step_end) of the grid owns
    int64_t shard_vals_end(int64_t step_end) const;
    void fill_kparams(KParams& kp, int64_t step_begin, int64_t step_end, int tile) const;

    // ---- Hessian (ctd_hess_host.cpp)
    HessModel H;
    void hess_step_blocks(int64_t i, std::vector<Block>& out) const;
    void hess_gen_column(int64_t j, std::vector<int64_t>& rows) const;   // rows >= j of column j, sorted
    int64_t hess_column_start(int64_t j) const;
    void fill_hparams(HParams& hp, int tile, int64_t step_begin = 0, int64_t step_end = 0) const;
};

// operator-level dependence masks (Model::dep_f / dep_g / dep_b) of the OCP functions, for CTD_PATTERN_OPTIMIZED
void compute_dep_masks(Model& m);
// builds Model::H (pattern bookkeeping + term tables); called by build_model
int build_hess_model(Model& m, std::string& err);
// position tables of the lane-per-step Hessian kernel (ctd_hess_step.hpp): pairs = (row, column) of the nout outputs of the
// step function; src[e] = output feeding position e of the segment (-1: structural zero), chunk_pos = positions per flush.
// false: the pattern does not hold every output in the order the step function produces them (caller keeps the tile kernel)
bool build_hess_step_tables(const Model& m, const short* pairs, int nout, int chunk, std::vector<int32_t>& src, std::vector<int32_t>& chunk_pos);
int default_hess_tile(const Model& m);

// status codes are those of include/ctdirect_hip.h; err receives a message on failure
int build_model(const HostDesc& d, Model& m, std::string& err);
struct InitSamples { int64_t n = 0; const double* t = nullptr; const double* state = nullptr; const double* control = nullptr; };
void model_initial_guess(const Model& m, double* x0, bool use_problem_default, const double* state, const double* control,
                         const double* variable, const InitSamples& samples = InitSamples{});
int default_tile(const Model& m, int64_t nsteps = 0);

}  // namespace ctd
