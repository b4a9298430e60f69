// ctd_jit.cpp -- run-time OCP registry: expression parser, functor generator (host only; the hiprtc side is in
// ctd_engine.hip).  See ctd_jit.hpp.
#include "ctd_jit.hpp"

#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <mutex>

#include "../../include/ctdirect_hip.h"
#include "ctd_hess.hpp"
#include "ctd_sym.hpp"

namespace ctd {

// ------------------------------------------------------------------------------------------------------
// expressions:  expr := term (('+'|'-') term)* ; term := unary (('*'|'/') unary)* ; unary := '-' unary | power ;
//               power := atom ('^' exponent)? ; exponent := ['-'] number | constant name | '(' constant expression ')'
//               atom := number | name | func '(' expr ')' | func2 '(' expr ',' expr ')' | '(' expr ')'
// names: t, x<k>, u<k>, v<k> (kind 0) or x0_<k>, xf_<k>, v<k> (kind 1), declared constants and aliases ("name = expression" entries
// of the constants string); functions exp log sin cos tan atan tanh sqrt abs asin acos sinh cosh floor, max(a, b) min(a, b)
// (derivatives of max / min: ForwardDiff's convention, d_gt in ctd_common.hpp; floor: zero)
// ------------------------------------------------------------------------------------------------------
namespace {
// common sub-expression pool of one generated function: every distinct non-constant function call (exp, sin, ...) is
// computed once into a temporary shared by all outputs of the function (hand-written functors do the same)
struct CsePool {
    std::vector<std::pair<std::string, std::string>> seen;   // (expression text, temporary)
    std::string decls;
    std::string intern(const std::string& code) {
        for (auto& e : seen) if (e.first == code) return e.second;
        const std::string name = "e" + std::to_string(seen.size());
        decls += "        const T " + name + " = " + code + ";\n";
        seen.emplace_back(code, name);
        return name;
    }
};

struct Parser {
    const std::string& s;
    const ExprCtx& cx;
    CsePool* cse = nullptr;
    size_t pos = 0;
    bool uses_t = false, uses_v = false;
    std::string err;
    RtProgram rpn;
    void op(uint8_t kind, int k = 0) { rpn.push_back(RtOp{kind, (int16_t)k}); }
    struct Val { std::string code; bool is_const; int node = -1; };
    // optional symbolic build (kind 0 expressions): the names t / x<k> / u<k> / v<k> are bound to these DAG nodes
    sym::Graph* g = nullptr;
    int g_t = -1;
    const int *g_x = nullptr, *g_u = nullptr, *g_v = nullptr;
    int depth = 0;                       // alias nesting

    Parser(const std::string& s_, const ExprCtx& cx_) : s(s_), cx(cx_) {}
    void skip() { while (pos < s.size() && std::isspace((unsigned char)s[pos])) ++pos; }
    bool fail(const std::string& m) { if (err.empty()) err = m + " at column " + std::to_string(pos + 1) + " of \"" + s + "\""; return false; }
    static std::string num(double v) {
        char buf[64];
        std::snprintf(buf, sizeof buf, "%.17g", v);
        std::string r(buf);
        if (r.find_first_of(".eEn") == std::string::npos) r += ".0";      // always a double literal
        return r;
    }
    bool index_of(const std::string& name, size_t prefix, int limit, int& k) {
        if (name.size() <= prefix) return false;
        for (size_t i = prefix; i < name.size(); ++i) if (!std::isdigit((unsigned char)name[i])) return false;
        k = std::atoi(name.c_str() + prefix);
        return k >= 1 && k <= limit;
    }
    bool expr(Val& out) {
        if (!term(out)) return false;
        for (;;) {
            skip();
            if (pos < s.size() && (s[pos] == '+' || s[pos] == '-')) {
                const char op = s[pos++];
                Val r;
                if (!term(r)) return false;
                out.code = "(" + out.code + " " + op + " " + r.code + ")";
                out.is_const = out.is_const && r.is_const;
                if (g) out.node = op == '+' ? g->add(out.node, r.node) : g->sub(out.node, r.node);
                this->op(op == '+' ? RT_ADD : RT_SUB);
            } else return true;
        }
    }
    bool term(Val& out) {
        if (!unary(out)) return false;
        for (;;) {
            skip();
            if (pos < s.size() && (s[pos] == '*' || s[pos] == '/')) {
                const char op = s[pos++];
                Val r;
                if (!unary(r)) return false;
                out.code = "(" + out.code + " " + op + " " + r.code + ")";
                out.is_const = out.is_const && r.is_const;
                if (g) out.node = op == '*' ? g->mul(out.node, r.node) : g->div(out.node, r.node);
                this->op(op == '*' ? RT_MUL : RT_DIV);
            } else return true;
        }
    }
    bool unary(Val& out) {
        skip();
        if (pos < s.size() && s[pos] == '-') {
            ++pos;
            Val r;
            if (!unary(r)) return false;
            out.code = "(-" + r.code + ")";
            out.is_const = r.is_const;
            if (g) out.node = g->neg(r.node);
            op(RT_NEG);
            return true;
        }
        if (pos < s.size() && s[pos] == '+') { ++pos; return unary(out); }
        return power(out);
    }
    // the exponent of '^': a constant -- a (signed) number, a declared constant, or a parenthesised constant expression.  Small
    // non-negative integers multiply out (d_powi, exact for every scalar type), anything else is a real power pow(x, p)
    bool exponent(double& p) {
        skip();
        double sign = 1.0;
        if (pos < s.size() && (s[pos] == '-' || s[pos] == '+')) { if (s[pos] == '-') sign = -1.0; ++pos; skip(); }
        if (pos >= s.size()) return fail("exponent expected after '^'");
        if (std::isdigit((unsigned char)s[pos]) || s[pos] == '.') {
            char* end = nullptr;
            p = sign * std::strtod(s.c_str() + pos, &end);
            if (end == s.c_str() + pos) return fail("malformed exponent");
            pos = (size_t)(end - s.c_str());
            return true;
        }
        // a constant name or a parenthesised constant expression: evaluated with a throw-away symbolic graph
        sym::Graph gc;
        Parser sub(s, cx);
        sub.pos = pos; sub.g = &gc; sub.depth = depth;
        static const int none[32] = {0};
        sub.g_t = gc.var(31); sub.g_x = sub.g_u = sub.g_v = none;
        Val v;
        if (!sub.atom(v)) { err = sub.err; return false; }
        if (!v.is_const || v.node < 0 || !gc.is_const(v.node)) return fail("the exponent of '^' must be a constant (write exp(b*log(a)) for a variable exponent)");
        p = sign * gc.cval(v.node);
        pos = sub.pos;
        return true;
    }
    bool power(Val& out) {
        if (!atom(out)) return false;
        skip();
        if (pos < s.size() && s[pos] == '^') {
            ++pos;
            double p = 0.0;
            if (!exponent(p)) return false;
            if (p == std::floor(p) && p >= 0.0 && p <= 64.0) {
                const int k = (int)p;
                if (k == 0) { out.code = "1.0"; out.is_const = true; }
                else if (k == 2) out.code = "d_sqr(" + out.code + ")";
                else if (k > 2) out.code = "d_powi(" + out.code + ", " + std::to_string(k) + ")";
                if (g) out.node = g->powi(out.node, k);
                op(RT_POW, k);
            } else {
                if (!(p == p) || std::fabs(p) > 1.0e6) return fail("exponent out of range");
                if (out.is_const) out.code = "::pow(" + out.code + ", " + num(p) + ")";
                else out.code = "d_powr(" + out.code + ", " + num(p) + ")";
                if (g) out.node = g->powr(out.node, p);
                if (cse && !out.is_const) out.code = cse->intern(out.code);
                op(RT_NONLIN);
            }
        }
        return true;
    }
    bool atom(Val& out) {
        skip();
        if (pos >= s.size()) return fail("unexpected end of expression");
        const char c = s[pos];
        if (c == '(') {
            ++pos;
            if (!expr(out)) return false;
            skip();
            if (pos >= s.size() || s[pos] != ')') return fail("')' expected");
            ++pos;
            return true;
        }
        if (std::isdigit((unsigned char)c) || c == '.') {
            char* end = nullptr;
            const double v = std::strtod(s.c_str() + pos, &end);
            if (end == s.c_str() + pos) return fail("malformed number");
            pos = (size_t)(end - s.c_str());
            out.code = num(v); out.is_const = true;
            if (g) out.node = g->constant(v);
            op(RT_CONST);
            return true;
        }
        if (std::isalpha((unsigned char)c) || c == '_') {
            size_t b = pos;
            while (pos < s.size() && (std::isalnum((unsigned char)s[pos]) || s[pos] == '_')) ++pos;
            const std::string name = s.substr(b, pos - b);
            skip();
            if (pos < s.size() && s[pos] == '(') {          // function call
                static const char* fn[][2] = {{"exp", "d_exp"}, {"sin", "d_sin"}, {"cos", "d_cos"}, {"sqrt", "d_sqrt"}, {"log", "d_log"},
                                              {"tan", "d_tan"}, {"atan", "d_atan"}, {"tanh", "d_tanh"}, {"abs", "d_abs"},
                                              {"asin", "d_asin"}, {"acos", "d_acos"}, {"sinh", "d_sinh"}, {"cosh", "d_cosh"}, {"floor", "d_floor"}};
                static const sym::Fn fid[] = {sym::F_EXP, sym::F_SIN, sym::F_COS, sym::F_SQRT, sym::F_LOG, sym::F_TAN, sym::F_ATAN, sym::F_TANH, sym::F_ABS,
                                              sym::F_ASIN, sym::F_ACOS, sym::F_SINH, sym::F_COSH, sym::F_FLOOR};
                if (name == "max" || name == "min") {        // two arguments; derivative = that of the selected operand (d_gt)
                    const bool mx = name == "max";
                    ++pos;
                    Val a, b;
                    if (!expr(a)) return false;
                    skip();
                    if (pos >= s.size() || s[pos] != ',') return fail("',' expected: " + name + " takes two arguments");
                    ++pos;
                    if (!expr(b)) return false;
                    skip();
                    if (pos >= s.size() || s[pos] != ')') return fail("')' expected");
                    ++pos;
                    out.is_const = a.is_const && b.is_const;
                    out.code = std::string(mx ? "d_max2<" : "d_min2<") + (out.is_const ? "double" : "T") + ">(" + a.code + ", " + b.code + ")";
                    if (g) out.node = mx ? g->max2(a.node, b.node) : g->min2(a.node, b.node);
                    if (cse && !out.is_const) out.code = cse->intern(out.code);
                    op(RT_MAX);
                    return true;
                }
                const char* target = nullptr;
                int fidx = -1;
                for (int i = 0; i < (int)(sizeof(fn) / sizeof(fn[0])); ++i) if (name == fn[i][0]) { target = fn[i][1]; fidx = i; }
                if (!target) return fail("unknown function '" + name + "' (available: exp, log, sin, cos, tan, atan, tanh, sqrt, abs, asin, acos, sinh, cosh, floor, max, min)");
                ++pos;
                Val a;
                if (!expr(a)) return false;
                skip();
                if (pos >= s.size() || s[pos] != ')') return fail("')' expected");
                ++pos;
                out.code = std::string(target) + "(" + a.code + ")";
                out.is_const = a.is_const;
                if (g) out.node = g->fn(fid[fidx], a.node);
                if (cse && !a.is_const) out.code = cse->intern(out.code);
                op(fid[fidx] == sym::F_FLOOR ? RT_ZERO : RT_NONLIN);
                return true;
            }
            int k = 0;
            out.is_const = false;
            if (cx.kind == 0) {
                if (name == "t") { uses_t = true; out.code = "t"; if (g) out.node = g_t; op(RT_T); return true; }
                if (name[0] == 'x' && index_of(name, 1, cx.n, k)) { out.code = "x[" + std::to_string(k - 1) + "]"; if (g) out.node = g_x[k - 1]; op(RT_X, k - 1); return true; }
                if (name[0] == 'u' && index_of(name, 1, cx.m, k)) { out.code = "u[" + std::to_string(k - 1) + "]"; if (g) out.node = g_u[k - 1]; op(RT_U, k - 1); return true; }
            } else {
                if (name.rfind("x0_", 0) == 0 && index_of(name, 3, cx.n, k)) { out.code = "x0[" + std::to_string(k - 1) + "]"; op(RT_X0, k - 1); return true; }
                if (name.rfind("xf_", 0) == 0 && index_of(name, 3, cx.n, k)) { out.code = "xf[" + std::to_string(k - 1) + "]"; op(RT_XF, k - 1); return true; }
            }
            if (name[0] == 'v' && index_of(name, 1, cx.nv, k)) { uses_v = true; out.code = "v[" + std::to_string(k - 1) + "]"; if (g) out.node = g_v[k - 1]; op(RT_V, k - 1); return true; }
            auto it = cx.constants.find(name);
            if (it != cx.constants.end()) { out.code = num(it->second); out.is_const = true; if (g) out.node = g->constant(it->second); op(RT_CONST); return true; }
            auto al = cx.aliases.find(name);
            if (al != cx.aliases.end()) {          // a named sub-expression: parsed in place (same graph, same temporaries, same postfix program)
                if (depth >= 24) return fail("aliases nested too deeply (or defined in terms of themselves): '" + name + "'");
                Parser sub(al->second, cx);
                sub.cse = cse; sub.g = g; sub.g_t = g_t; sub.g_x = g_x; sub.g_u = g_u; sub.g_v = g_v; sub.depth = depth + 1;
                Val v;
                if (!sub.expr(v)) { err = "in alias '" + name + "': " + sub.err; return false; }
                sub.skip();
                if (sub.pos != al->second.size()) { sub.fail("unexpected trailing input"); err = "in alias '" + name + "': " + sub.err; return false; }
                uses_t = uses_t || sub.uses_t; uses_v = uses_v || sub.uses_v;
                rpn.insert(rpn.end(), sub.rpn.begin(), sub.rpn.end());
                out = v;
                out.code = "(" + v.code + ")";
                return true;
            }
            return fail("unknown name '" + name + "'");
        }
        return fail(std::string("unexpected character '") + c + "'");
    }
};

bool parse_constants(const char* text, std::map<std::string, double>& out, std::map<std::string, std::string>& aliases, std::string& err) {
    out.clear(); aliases.clear();
    if (!text) return true;
    std::string s(text);
    size_t pos = 0;
    while (pos < s.size()) {
        size_t end = s.find(';', pos);
        if (end == std::string::npos) end = s.size();
        std::string item = s.substr(pos, end - pos);
        pos = end + 1;
        size_t eq = item.find('=');
        auto trim = [](std::string v) {
            size_t a = v.find_first_not_of(" \t\n"), b = v.find_last_not_of(" \t\n");
            return a == std::string::npos ? std::string() : v.substr(a, b - a + 1);
        };
        if (trim(item).empty()) continue;
        if (eq == std::string::npos) { err = "constant '" + item + "' needs the form name=value"; return false; }
        const std::string name = trim(item.substr(0, eq)), val = trim(item.substr(eq + 1));
        if (name.empty() || !(std::isalpha((unsigned char)name[0]) || name[0] == '_')) { err = "bad constant name '" + name + "'"; return false; }
        for (char ch : name) if (!(std::isalnum((unsigned char)ch) || ch == '_')) { err = "bad constant name '" + name + "'"; return false; }
        char* e = nullptr;
        const double v = std::strtod(val.c_str(), &e);
        if (val.empty()) { err = "constant '" + name + "' has no value"; return false; }
        if (out.count(name) || aliases.count(name)) { err = "'" + name + "' is defined twice"; return false; }
        if (*e != 0) {                    // not a plain number: a named sub-expression (checked where it is used)
            if (val.size() > 20000) { err = "alias '" + name + "' is too long"; return false; }
            aliases[name] = val;
            continue;
        }
        out[name] = v;
    }
    return true;
}

std::mutex g_mu;
std::vector<std::unique_ptr<RtOcp>> g_ocps;
}  // namespace

static bool expr_to_cpp_cse(const std::string& expr, const ExprCtx& cx, std::string& out, bool& is_const, bool& uses_t, bool& uses_v,
                            std::string& err, RtProgram* prog, CsePool* cse);

bool expr_to_cpp(const std::string& expr, const ExprCtx& cx, std::string& out, bool& is_const, bool& uses_t, bool& uses_v,
                 std::string& err, RtProgram* prog) {
    return expr_to_cpp_cse(expr, cx, out, is_const, uses_t, uses_v, err, prog, nullptr);
}

static bool expr_to_cpp_cse(const std::string& expr, const ExprCtx& cx, std::string& out, bool& is_const, bool& uses_t, bool& uses_v,
                            std::string& err, RtProgram* prog, CsePool* cse) {
    Parser p(expr, cx);
    p.cse = cse;
    Parser::Val v;
    if (!p.expr(v)) { err = p.err; return false; }
    p.skip();
    if (p.pos != expr.size()) { p.fail("unexpected trailing input"); err = p.err; return false; }
    out = v.code; is_const = v.is_const; uses_t = p.uses_t; uses_v = p.uses_v;
    if (prog) *prog = p.rpn;
    return true;
}

const RtOcp* runtime_ocp(int id) {
    std::lock_guard<std::mutex> lk(g_mu);
    const int k = id - kRuntimeIdBase;
    return (k >= 0 && k < (int)g_ocps.size()) ? g_ocps[k].get() : nullptr;
}

// Symbolic second derivatives of the scalar a stage-type point contributes to the Lagrangian (ctd_sym.hpp, SymPrm in
// ctd_hess.hpp): returns the bodies of UserOCP::stage_sym_irk / stage_sym_mid, which write the md x md upper triangle of the
// point's record (and, for Gauss-Legendre stages with free times, the RK helper block behind it).
static bool gen_sym_stage(const ctd_ocp_def* d, const ExprCtx& c0, bool has_lag, int kind, std::string& body, std::string& err) {
    const bool irk = kind == 0, trap = kind == 2;
    const int n = d->n, m = d->m, nv = d->nv, md = n + m + nv;
    const SymPrm P = sym_prm(n, m, nv, d->npath);
    const bool free_time = d->it0 >= 0 || d->itf >= 0;
    sym::Graph g;
    std::vector<int> X(n), U(m > 0 ? m : 1), V(nv > 0 ? nv : 1);
    int t = g.param(P.T0), h = g.param(P.H0);
    for (int k = 0; k < nv; ++k) {
        t = g.add(t, g.mul(g.param(P.TD + k), g.var(n + m + k)));
        h = g.add(h, g.mul(g.param(P.HD + k), g.var(n + m + k)));
    }
    for (int r = 0; r < n; ++r) {
        int x = g.add(g.param(P.X0 + r), g.var(r));
        if (irk) for (int k = 0; k < nv; ++k) x = g.add(x, g.mul(g.mul(g.param(P.HD + k), g.param(P.KAP + r)), g.var(n + m + k)));
        X[r] = x;
    }
    for (int b = 0; b < m; ++b) U[b] = g.add(g.param(P.U0 + b), g.var(n + b));
    for (int k = 0; k < nv; ++k) V[k] = g.add(g.param(P.V0 + k), g.var(n + m + k));
    auto parse = [&](const char* text, int& node) {
        const std::string str(text);
        Parser ps(str, c0);
        ps.g = &g; ps.g_t = t; ps.g_x = X.data(); ps.g_u = U.data(); ps.g_v = V.data();
        Parser::Val v;
        if (!ps.expr(v)) { err = ps.err; return false; }
        node = v.node;
        return true;
    };
    int hm = g.param(P.HM0);
    for (int k = 0; k < nv; ++k) hm = g.add(hm, g.mul(g.param(P.HMD + k), g.var(n + m + k)));
    int F = g.constant(0.0);
    for (int r = 0; r < n; ++r) {
        int fr = -1;
        if (!parse(d->dynamics[r], fr)) return false;
        const int w = trap ? g.mul(g.constant(-0.5), g.add(g.mul(hm, g.param(P.WP + r)), g.mul(h, g.param(P.W + r)))) : g.param(P.W + r);
        F = g.add(F, g.mul(w, fr));
    }
    int L = g.constant(0.0);
    if (has_lag && !parse(d->lagrange, L)) return false;
    int Phi;
    if (irk) Phi = g.add(F, g.mul(g.param(P.CL), g.mul(h, L)));
    else if (!trap) Phi = g.mul(h, g.add(F, g.mul(g.param(P.CL), L)));
    else {
        Phi = g.add(F, g.mul(g.param(P.CL), g.mul(g.add(hm, h), L)));
        for (int r = 0; r < d->npath; ++r) {
            int gr = -1;
            if (!parse(d->path[r], gr)) return false;
            Phi = g.add(Phi, g.mul(g.param(P.WG + r), gr));
        }
    }
    std::vector<std::pair<std::string, int>> outs;
    std::vector<int> d1(md);
    for (int p = 0; p < md; ++p) d1[p] = g.diff(Phi, p);
    for (int p = 0; p < md; ++p)
        for (int q = p; q < md; ++q)
            outs.emplace_back("HD[" + std::to_string(hess_tri(md, p, q)) + "]", g.at_zero(g.diff(d1[p], q)));
    if (irk && free_time) {      // RK[k][a] = h d2Phi/dx_a dV_k + dh/dv_k dPhi/dx_a   (hess_eval_stage)
        const int oRK = hess_tri_size(md);
        for (int k = 0; k < nv; ++k)
            for (int a = 0; a < n; ++a)
                outs.emplace_back("HD[" + std::to_string(oRK + k * n + a) + "]",
                                  g.add(g.mul(g.param(P.H0), g.at_zero(g.diff(d1[a], n + m + k))),
                                        g.mul(g.param(P.HD + k), g.at_zero(d1[a]))));
    }
    if (g.nodes.size() > 200000) { err = "symbolic derivatives too large"; return false; }
    body = g.codegen(outs, "p", "        ");
    return true;
}

// Symbolic second derivatives of the path point  sum_r WG_r g_r(t, x, u, v)  (SymPathPrm in ctd_hess.hpp) along the record's
// directions (x | u | v, the v directions including the motion of t): body of SymPathH<P>::eval(p, HP), HP = packed triangle.
// Every entry is written (structural zeros as 0.0: the host's dependency probe may keep terms on entries that vanish identically).
static bool gen_sym_pathh(const ctd_ocp_def* d, const ExprCtx& c0, std::string& body, std::string& err) {
    const int n = d->n, m = d->m, nv = d->nv, np = d->npath, md = n + m + nv;
    const SymPathPrm P = sym_path_prm(n, m, nv, np);
    sym::Graph g;
    std::vector<int> X(n), U(m > 0 ? m : 1), V(nv > 0 ? nv : 1);
    int t = g.param(P.T0);
    for (int k = 0; k < nv; ++k) t = g.add(t, g.mul(g.param(P.TD + k), g.var(n + m + k)));
    for (int r = 0; r < n; ++r) X[r] = g.add(g.param(P.X0 + r), g.var(r));
    for (int b = 0; b < m; ++b) U[b] = g.add(g.param(P.U0 + b), g.var(n + b));
    for (int k = 0; k < nv; ++k) V[k] = g.add(g.param(P.V0 + k), g.var(n + m + k));
    int Phi = g.constant(0.0);
    for (int r = 0; r < np; ++r) {
        const std::string str(d->path[r]);
        Parser ps(str, c0);
        ps.g = &g; ps.g_t = t; ps.g_x = X.data(); ps.g_u = U.data(); ps.g_v = V.data();
        Parser::Val v;
        if (!ps.expr(v)) { err = ps.err; return false; }
        Phi = g.add(Phi, g.mul(g.param(P.WG + r), v.node));
    }
    std::vector<std::pair<std::string, int>> outs;
    for (int p = 0; p < md; ++p) {
        const int d1 = g.diff(Phi, p);
        for (int q = p; q < md; ++q) {
            outs.emplace_back("HP[" + std::to_string(hess_tri(md, p, q)) + "]", g.at_zero(g.diff(d1, q)));
        }
    }
    body = outs.empty() ? std::string() : g.codegen(outs, "p", "        ");
    return true;
}

// First derivatives of the dynamics at an evaluation point, written straight into an eval block of the step record of the
// constraint / Jacobian kernel (ctd_layout.hpp: F[n x ldx] | G[n x ldu] | W[n x nv] | f[n] | ft[n]): body of
// UserOCP::dyn_sym(t, x, u, v, ev).  Every entry of F and G is written (structural zeros as 0.0); ft / W only when the
// dynamics depend on t / v explicitly, as eval_dynamics does.
// `nparts` > 1: body = the cases of a switch on `part`; part k holds the outputs of the rows r = k, k + nparts, ... (the rows
// of an OCP come in classes of similar cost -- kinematics, rotation, inertia -- so striding balances the parts); the lanes of
// one evaluation point that would each differentiate a chunk of directions with duals split the symbolic code this way
// nz (optional): the structural nonzeros of df/dx and df/du -- map_f[r n + c] / map_g[r m + c] = slot in row-major order or -1.
// The code stores the nonzeros only, at the offsets of the SPARSE eval block (make_rec_layout with nF, nG; DynNZ in
// ctd_kernel_body.hpp); CTD_DENSE_EVAL=1 at generation time keeps the dense n x ldx / n x ldu blocks (every zero stored).
static bool gen_sym_dyn(const ctd_ocp_def* d, const ExprCtx& c0, bool& dyn_t, bool& dyn_v, std::string& body, std::string& err, int nparts = 1,
                        DynNZMap* nz = nullptr, size_t* code_lines = nullptr) {
    const int n = d->n, m = d->m, nv = d->nv;
    sym::Graph g;
    // parameters: p[0] = t, then x[n], u[m], v[nv]; variables: the same positions
    std::vector<int> X(n), U(m > 0 ? m : 1), V(nv > 0 ? nv : 1);
    const int t = g.add(g.param(0), g.var(0));
    for (int r = 0; r < n; ++r) X[r] = g.add(g.param(1 + r), g.var(1 + r));
    for (int b = 0; b < m; ++b) U[b] = g.add(g.param(1 + n + b), g.var(1 + n + b));
    for (int k = 0; k < nv; ++k) V[k] = g.add(g.param(1 + n + m + k), g.var(1 + n + m + k));
    std::vector<int> f(n);
    dyn_t = dyn_v = false;
    for (int r = 0; r < n; ++r) {
        const std::string str(d->dynamics[r]);
        Parser ps(str, c0);
        ps.g = &g; ps.g_t = t; ps.g_x = X.data(); ps.g_u = U.data(); ps.g_v = V.data();
        Parser::Val v;
        if (!ps.expr(v)) { err = ps.err; return false; }
        f[r] = v.node;
        dyn_t = dyn_t || ps.uses_t; dyn_v = dyn_v || ps.uses_v;
    }
    // first partials, and which of them vanish identically
    std::vector<int> dfx(n * n), dfu(n * (m > 0 ? m : 1));
    DynNZMap map;
    const char* dense_env = std::getenv("CTD_DENSE_EVAL");
    map.sparse = !(dense_env && std::atoi(dense_env) != 0);
    map.n_f = map.n_g = 0;
    map.map_f.assign(n * n, -1); map.map_g.assign(n * m, -1);
    for (int r = 0; r < n; ++r) {
        for (int c = 0; c < n; ++c) { dfx[r * n + c] = g.at_zero(g.diff(f[r], 1 + c)); if (!g.is_zero(dfx[r * n + c])) map.map_f[r * n + c] = map.n_f++; }
        for (int b = 0; b < m; ++b) { dfu[r * m + b] = g.at_zero(g.diff(f[r], 1 + n + b)); if (!g.is_zero(dfu[r * m + b])) map.map_g[r * m + b] = map.n_g++; }
    }
    const RecLayout R = map.sparse ? make_rec_layout(n, m, nv, d->npath, d->nbc, 0, n + d->npath, -1, map.n_f, map.n_g)
                                   : make_rec_layout(n, m, nv, d->npath, d->nbc, 0, n + d->npath);     // eval-block offsets depend on n, m, nv (and the nonzeros) only
    if (nz) *nz = map;
    body.clear();
    for (int part = 0; part < nparts; ++part) {
        std::vector<std::pair<std::string, int>> outs;
        for (int r = part; r < n; r += nparts) {
            for (int c = 0; c < n; ++c) {
                if (!map.sparse) outs.emplace_back("ev[" + std::to_string(R.oF + r * R.ldx + c) + "]", dfx[r * n + c]);
                else if (map.map_f[r * n + c] >= 0) outs.emplace_back("ev[" + std::to_string(R.oF + map.map_f[r * n + c]) + "]", dfx[r * n + c]);
            }
            for (int b = 0; b < m; ++b) {
                if (!map.sparse) outs.emplace_back("ev[" + std::to_string(R.oG + r * R.ldu + b) + "]", dfu[r * m + b]);
                else if (map.map_g[r * m + b] >= 0) outs.emplace_back("ev[" + std::to_string(R.oG + map.map_g[r * m + b]) + "]", dfu[r * m + b]);
            }
            if (dyn_t) outs.emplace_back("ev[" + std::to_string(R.oft + r) + "]", g.at_zero(g.diff(f[r], 0)));
            if (dyn_v) for (int k = 0; k < nv; ++k) outs.emplace_back("ev[" + std::to_string(R.oW + r * nv + k) + "]", g.at_zero(g.diff(f[r], 1 + n + m + k)));
            outs.emplace_back("ev[" + std::to_string(R.of + r) + "]", g.at_zero(f[r]));
        }
        if (nparts == 1) body = g.codegen(outs, "p", "        ");
        else body += "            case " + std::to_string(part) + ": {\n" + g.codegen(outs, "p", "                ") + "            } break;\n";
    }
    if (code_lines) { *code_lines = 0; for (char ch : body) if (ch == '\n') ++*code_lines; }
    return true;
}

std::string dyn_nz_source(const std::string& type, int n, int m, const DynNZMap& nz) {
    auto table = [](const std::vector<int>& t) {
        std::string s;
        for (size_t k = 0; k < t.size(); ++k) s += (k ? ", " : "") + std::to_string(t[k]);
        return t.empty() ? std::string("-1") : s;
    };
    std::string s = "template <> struct DynNZ<" + type + "> {\n    static constexpr bool sparse = true;\n";
    s += "    static constexpr int nF = " + std::to_string(nz.n_f) + ", nG = " + std::to_string(nz.n_g) + ";\n";
    s += "    CTD_HD static constexpr int fx(int r, int c) { constexpr short t[] = {" + table(nz.map_f) + "}; return t[r * " + std::to_string(n) + " + c]; }\n";
    s += "    CTD_HD static constexpr int gu(int r, int c) { constexpr short t[] = {" + table(nz.map_g) + "}; return t[r * " + std::to_string(m > 0 ? m : 1) + " + c]; }\n";
    s += "};\n";
    return s;
}

// The same for the path constraints g(t, x, u, v): Px[np x ldx] | Pu[np x ldu] | Pv[np x nv] | Pt[np] of the step record and
// the values into `val` (eval_path): body of UserOCP::path_sym(p, rec, val).
static bool gen_sym_path(const ctd_ocp_def* d, const ExprCtx& c0, std::string& body, std::string& err) {
    const int n = d->n, m = d->m, nv = d->nv, np = d->npath;
    const int ldx = n | 1, ldu = m | 1;
    sym::Graph g;
    std::vector<int> X(n), U(m > 0 ? m : 1), V(nv > 0 ? nv : 1);
    const int t = g.add(g.param(0), g.var(0));
    for (int r = 0; r < n; ++r) X[r] = g.add(g.param(1 + r), g.var(1 + r));
    for (int b = 0; b < m; ++b) U[b] = g.add(g.param(1 + n + b), g.var(1 + n + b));
    for (int k = 0; k < nv; ++k) V[k] = g.add(g.param(1 + n + m + k), g.var(1 + n + m + k));
    std::vector<int> f(np > 0 ? np : 1);
    bool pt = false, pv = false;
    for (int r = 0; r < np; ++r) {
        const std::string str(d->path[r]);
        Parser ps(str, c0);
        ps.g = &g; ps.g_t = t; ps.g_x = X.data(); ps.g_u = U.data(); ps.g_v = V.data();
        Parser::Val v;
        if (!ps.expr(v)) { err = ps.err; return false; }
        f[r] = v.node;
        pt = pt || ps.uses_t; pv = pv || ps.uses_v;
    }
    // offsets relative to oPx (the caller passes rec + R.oPx): Px | Pu | Pv | Pt, as in make_rec_layout
    const int oPu = np * ldx, oPv = oPu + np * ldu, oPt = oPv + np * nv;
    std::vector<std::pair<std::string, int>> outs;
    for (int r = 0; r < np; ++r) {
        for (int c = 0; c < n; ++c) outs.emplace_back("px[" + std::to_string(r * ldx + c) + "]", g.at_zero(g.diff(f[r], 1 + c)));
        for (int b = 0; b < m; ++b) outs.emplace_back("px[" + std::to_string(oPu + r * ldu + b) + "]", g.at_zero(g.diff(f[r], 1 + n + b)));
        if (pt) outs.emplace_back("px[" + std::to_string(oPt + r) + "]", g.at_zero(g.diff(f[r], 0)));
        if (pv) for (int k = 0; k < nv; ++k) outs.emplace_back("px[" + std::to_string(oPv + r * nv + k) + "]", g.at_zero(g.diff(f[r], 1 + n + m + k)));
        outs.emplace_back("val[" + std::to_string(r) + "]", g.at_zero(f[r]));
    }
    body = g.codegen(outs, "p", "        ");
    return true;
}

// Value and first derivatives of the Lagrange cost l(t, x, u, v) for the gradient kernel (lagrange_partials, ctd_kernels.hpp):
// body of UserOCP::lag_sym(p, out), out = [value | l_x[n] | l_u[m] | l_t | l_v[nv]].
static bool gen_sym_lag(const ctd_ocp_def* d, const ExprCtx& c0, std::string& body, std::string& err) {
    const int n = d->n, m = d->m, nv = d->nv;
    sym::Graph g;
    std::vector<int> X(n), U(m > 0 ? m : 1), V(nv > 0 ? nv : 1);
    const int t = g.add(g.param(0), g.var(0));
    for (int r = 0; r < n; ++r) X[r] = g.add(g.param(1 + r), g.var(1 + r));
    for (int b = 0; b < m; ++b) U[b] = g.add(g.param(1 + n + b), g.var(1 + n + b));
    for (int k = 0; k < nv; ++k) V[k] = g.add(g.param(1 + n + m + k), g.var(1 + n + m + k));
    const std::string str(d->lagrange);
    Parser ps(str, c0);
    ps.g = &g; ps.g_t = t; ps.g_x = X.data(); ps.g_u = U.data(); ps.g_v = V.data();
    Parser::Val v;
    if (!ps.expr(v)) { err = ps.err; return false; }
    std::vector<std::pair<std::string, int>> outs;
    outs.emplace_back("out[0]", g.at_zero(v.node));
    for (int c = 0; c < n; ++c) outs.emplace_back("out[" + std::to_string(1 + c) + "]", g.at_zero(g.diff(v.node, 1 + c)));
    for (int b = 0; b < m; ++b) outs.emplace_back("out[" + std::to_string(1 + n + b) + "]", g.at_zero(g.diff(v.node, 1 + n + b)));
    outs.emplace_back("out[" + std::to_string(1 + n + m) + "]", g.at_zero(g.diff(v.node, 0)));
    for (int k = 0; k < nv; ++k) outs.emplace_back("out[" + std::to_string(2 + n + m + k) + "]", g.at_zero(g.diff(v.node, 1 + n + m + k)));
    body = g.codegen(outs, "p", "        ");
    return true;
}

int register_runtime_ocp(const ctd_ocp_def* d, int* id, std::string& err) {
    if (!d || !id) { err = "null argument"; return CTD_EINVAL; }
    if (d->n < 1 || d->n > 24 || d->m < 0 || d->m > 12 || d->nv < 0 || d->nv > kMaxNV || d->npath < 0 || d->npath > 16 ||
        d->nbc < 0 || d->nbc > 64) { err = "dimensions out of range (1 <= n <= 24, m <= 12, nv <= 4, npath <= 16, nbc <= 64)"; return CTD_EINVAL; }
    if (2 * d->n + d->nv > 31 || d->n + d->m + d->nv > 31) { err = "too many differentiation directions (n + m + nv and 2 n + nv must be <= 31)"; return CTD_EINVAL; }
    if (d->it0 >= d->nv || d->itf >= d->nv || d->it0 < -1 || d->itf < -1 || (d->it0 >= 0 && d->it0 == d->itf)) { err = "it0 / itf must be -1 or distinct indices into v"; return CTD_EINVAL; }
    if (!d->dynamics) { err = "dynamics expressions missing"; return CTD_EINVAL; }
    if ((d->npath > 0 && !d->path) || (d->nbc > 0 && !d->boundary)) { err = "path / boundary expressions missing"; return CTD_EINVAL; }
    auto o = std::make_unique<RtOcp>();
    o->name = d->name ? d->name : "user_ocp";
    // the name goes into a comment of the generated source: only [A-Za-z0-9_.-] (anything else -- a newline above all --
    // could smuggle text into the compiled module; the ABI accepts expressions, never C++)
    if (o->name.empty() || o->name.size() > 64) { err = "problem name must have 1..64 characters"; return CTD_EINVAL; }
    for (const char ch : o->name)
        if (!((ch >= 'A' && ch <= 'Z') || (ch >= 'a' && ch <= 'z') || (ch >= '0' && ch <= '9') || ch == '_' || ch == '.' || ch == '-')) {
            err = "problem name may only hold the characters A-Z a-z 0-9 _ . -";
            return CTD_EINVAL;
        }
    ExprCtx c0{d->n, d->m, d->nv, 0, {}, {}}, c1{d->n, d->m, d->nv, 1, {}, {}};
    if (!parse_constants(d->constants, c0.constants, c0.aliases, err)) return CTD_EINVAL;
    c1.constants = c0.constants; c1.aliases = c0.aliases;
    for (const auto& al : c0.aliases)         // names the grammar reserves cannot be aliased (x1, u2, t, ... would shadow the variables)
        if (al.first == "t" || ((al.first[0] == 'x' || al.first[0] == 'u' || al.first[0] == 'v') && al.first.size() > 1 && std::isdigit((unsigned char)al.first[1]))) {
            err = "alias name '" + al.first + "' collides with a variable name";
            return CTD_EINVAL;
        }
    o->dyn_t = o->dyn_v = o->path_t = o->path_v = o->lag_t = o->lag_v = false;
    std::string body_dyn, body_lag, body_may, body_path, body_bnd;
    CsePool pool_dyn, pool_lag, pool_may, pool_path, pool_bnd;
    auto emit = [&](const char* text, const ExprCtx& cx, const std::string& lhs, std::string& body, bool& ut, bool& uv, const char* what, int idx,
                    RtProgram& prog, CsePool& pool) {
        if (!text) { err = std::string(what) + " expression " + std::to_string(idx + 1) + " is null"; return false; }
        std::string code, e;
        bool isc = false, t_ = false, v_ = false;
        if (!expr_to_cpp_cse(text, cx, code, isc, t_, v_, e, &prog, &pool)) { err = std::string(what) + " " + std::to_string(idx + 1) + ": " + e; return false; }
        ut = ut || t_; uv = uv || v_;
        body += "        " + lhs + (isc ? " T(" + code + ");\n" : " " + code + ";\n");
        return true;
    };
    bool dummy_t = false, dummy_v = false;
    o->p_dynamics.resize(d->n); o->p_path.resize(d->npath); o->p_boundary.resize(d->nbc);
    for (int r = 0; r < d->n; ++r)
        if (!emit(d->dynamics[r], c0, "dx[" + std::to_string(r) + "] =", body_dyn, o->dyn_t, o->dyn_v, "dynamics", r, o->p_dynamics[r], pool_dyn)) return CTD_EINVAL;
    const bool has_lag = d->lagrange && *d->lagrange, has_may = d->mayer && *d->mayer;
    if (has_lag && !emit(d->lagrange, c0, "return", body_lag, o->lag_t, o->lag_v, "lagrange", 0, o->p_lagrange, pool_lag)) return CTD_EINVAL;
    if (has_may && !emit(d->mayer, c1, "return", body_may, dummy_t, dummy_v, "mayer", 0, o->p_mayer, pool_may)) return CTD_EINVAL;
    for (int r = 0; r < d->npath; ++r)
        if (!emit(d->path[r], c0, "r[" + std::to_string(r) + "] =", body_path, o->path_t, o->path_v, "path", r, o->p_path[r], pool_path)) return CTD_EINVAL;
    for (int r = 0; r < d->nbc; ++r)
        if (!emit(d->boundary[r], c1, "r[" + std::to_string(r) + "] =", body_bnd, dummy_t, dummy_v, "boundary", r, o->p_boundary[r], pool_bnd)) return CTD_EINVAL;
    if (!has_lag) body_lag = "        return T(0.0);\n";
    if (!has_may) body_may = "        return T(0.0);\n";
    o->dc = 4;
    o->hk = d->n >= 8 ? 1 : 4;          // ctd::HessK (ctd_hess_body.hpp)
    o->maxb = 256;
    auto B = [](bool b) { return b ? "true" : "false"; };
    std::string& s = o->functor_src;
    s = "namespace ctd {\n// generated by ctd_register_ocp from the expressions of '" + o->name + "'\nstruct UserOCP {\n";
    s += "    static constexpr int NX = " + std::to_string(d->n) + ", NU = " + std::to_string(d->m) + ", NV = " + std::to_string(d->nv) +
         ", NPATH = " + std::to_string(d->npath) + ", NBC = " + std::to_string(d->nbc) + ";\n";
    s += "    static constexpr int IT0 = " + std::to_string(d->it0) + ", ITF = " + std::to_string(d->itf) + ";\n";
    s += std::string("    static constexpr bool HAS_LAGRANGE = ") + B(has_lag) + ", HAS_MAYER = " + B(has_may) + ";\n";
    s += std::string("    static constexpr bool DYN_T = ") + B(o->dyn_t) + ", DYN_V = " + B(o->dyn_v) + ", PATH_T = " + B(o->path_t) +
         ", PATH_V = " + B(o->path_v) + ", LAG_T = " + B(o->lag_t) + ", LAG_V = " + B(o->lag_v) + ";\n";
    s += "    static constexpr int DC = " + std::to_string(o->dc) + ", MAXB = " + std::to_string(o->maxb) + ";\n";
    s += "    template <class T> CTD_HD static void dynamics(T* dx, const T& t, const T* x, const T* u, const T* v) {\n" + pool_dyn.decls + body_dyn + "    }\n";
    s += "    template <class T> CTD_HD static T lagrange(const T& t, const T* x, const T* u, const T* v) {\n" + pool_lag.decls + body_lag + "    }\n";
    s += "    template <class T> CTD_HD static T mayer(const T* x0, const T* xf, const T* v) {\n" + pool_may.decls + body_may + "    }\n";
    s += "    template <class T> CTD_HD static void path(T* r, const T& t, const T* x, const T* u, const T* v) {\n" + pool_path.decls + body_path + "    }\n";
    s += "    template <class T> CTD_HD static void boundary(T* r, const T* x0, const T* xf, const T* v) {\n" + pool_bnd.decls + body_bnd + "    }\n";
    // symbolically differentiated stage functions for the Hessian kernel (opt-out: CTD_HESS_SYM=0 at registration)
    {
        const char* env = std::getenv("CTD_HESS_SYM");
        std::string b_irk, b_mid, b_trap, e2;
        o->has_sym = !(env && std::string(env) == "0") && gen_sym_stage(d, c0, has_lag, 0, b_irk, e2) &&
                     gen_sym_stage(d, c0, has_lag, 1, b_mid, e2) && gen_sym_stage(d, c0, has_lag, 2, b_trap, e2);
        s += std::string("    static constexpr bool HAS_SYM = ") + B(o->has_sym) + ";\n";
        if (o->has_sym) {
            s += "    CTD_HD static void stage_sym_irk(const double* p, double* HD) {\n" + b_irk + "    }\n";
            s += "    CTD_HD static void stage_sym_mid(const double* p, double* HD) {\n" + b_mid + "    }\n";
            s += "    CTD_HD static void stage_sym_trap(const double* p, double* HD) {\n" + b_trap + "    }\n";
        }
    }
    {   // symbolic first derivatives of the dynamics for the constraint / Jacobian kernel (opt-out: CTD_DYN_SYM=0)
        const char* env = std::getenv("CTD_DYN_SYM");
        std::string b_dyn, e2;
        bool dt = false, dv = false;
        DynNZMap nz;
        size_t dyn_lines = 0;
        bool ok = !(env && std::string(env) == "0") && gen_sym_dyn(d, c0, dt, dv, b_dyn, e2, 1, &nz, &dyn_lines);
        // LONG generated code (the reference's swimmer: 213 statements with ~250 trigonometric terms per evaluation point; the 12-state
        // quadrotor has 68) stays on forward duals in the constraint / Jacobian kernel.  Measured on MI355X, ROCm 7.2
        // (profiles/r04_experiments.md section 5): the kernel with that code inlined was FRAGILE -- compiled for four waves per SIMD it
        // spilled 540 registers and computed garbage on every lane but the first of a divergent wave; for one wave per SIMD it was right,
        // wrong or faulting (memory aperture violation) depending on unrelated small changes elsewhere in the kernel source -- while the
        // same phase functions with the same functor run clean under AddressSanitizer / UBSan in emulation (tests/emu, emu_user_cons_jac)
        // and the dual-number kernel was exact in every build.  CTD_JIT_LONG_SYM=1 keeps the generated code (experiments).
        if (ok && dyn_lines > 120 && !std::getenv("CTD_JIT_LONG_SYM")) ok = false;
        if (ok) o->dyn_nz = nz;
        s += std::string("    static constexpr bool HAS_SYM_DYN = ") + B(ok) + ";\n";
        if (ok) s += "    CTD_HD static void dyn_sym(const double* p, double* ev) {\n" + b_dyn + "    }\n";
        // wide OCPs (four direction chunks and more, Dirs<P>::NCH_DYN): the same code split by rows, one part per wave (SymDyn::parts)
        const int ndir = d->n + d->m + (dt ? 1 : 0) + (dv ? d->nv : 0), nparts = (ndir + o->dc - 1) / o->dc;
        std::string b_parts;
        bool dt2 = false, dv2 = false;
        const bool okp4 = ok && nparts >= 4 && gen_sym_dyn(d, c0, dt2, dv2, b_parts, e2, nparts);
        s += "    static constexpr int DYN_PARTS = " + std::to_string(okp4 ? nparts : 1) + ";\n";
        if (okp4) s += "    CTD_HD static void dyn_sym_part(int part, const double* p, double* ev) {\n        switch (part) {\n" + b_parts + "        }\n    }\n";
        std::string b_path;
        const bool okp = ok && d->npath > 0 && gen_sym_path(d, c0, b_path, e2);
        std::string b_lag;
        const bool okl = ok && has_lag && gen_sym_lag(d, c0, b_lag, e2);
        s += std::string("    static constexpr bool HAS_SYM_LAG = ") + B(okl) + ";\n";
        if (okl) s += "    CTD_HD static void lag_sym(const double* p, double* out) {\n" + b_lag + "    }\n";
        s += std::string("    static constexpr bool HAS_SYM_PATH = ") + B(okp) + ";\n";
        if (okp) s += "    CTD_HD static void path_sym(const double* p, double* px, double* val) {\n" + b_path + "    }\n";
    }
    s += "};\n";
    if (o->dyn_nz.sparse) s += dyn_nz_source("UserOCP", d->n, d->m, o->dyn_nz);
    s += "}  // namespace ctd\n";

    ProblemInfo& pi = o->info;
    pi.name = o->name.c_str();
    pi.n = d->n; pi.m = d->m; pi.nv = d->nv; pi.npath = d->npath; pi.nbc = d->nbc;
    pi.it0 = d->it0; pi.itf = d->itf; pi.t0 = d->t0; pi.tf = d->tf;
    pi.lagrange = has_lag; pi.mayer = has_may; pi.maximize = d->maximize != 0;
    auto box = [](int dim, const double* lb, const double* ub, std::vector<BoxItem>& out) {
        out.clear();
        for (int k = 0; k < dim; ++k) {
            const double l = lb ? lb[k] : -kInf, u = ub ? ub[k] : kInf;
            if (l > -kInf || u < kInf) out.push_back(BoxItem{k, l, u});
        }
    };
    box(d->n, d->state_lb, d->state_ub, pi.state_box);
    box(d->m, d->control_lb, d->control_ub, pi.control_box);
    box(d->nv, d->variable_lb, d->variable_ub, pi.variable_box);
    pi.path_lb.assign(d->npath, 0.0); pi.path_ub.assign(d->npath, 0.0);
    pi.bc_lb.assign(d->nbc, 0.0); pi.bc_ub.assign(d->nbc, 0.0);
    for (int k = 0; k < d->npath; ++k) { if (d->path_lb) pi.path_lb[k] = d->path_lb[k]; if (d->path_ub) pi.path_ub[k] = d->path_ub[k]; }
    for (int k = 0; k < d->nbc; ++k) { if (d->boundary_lb) pi.bc_lb[k] = d->boundary_lb[k]; if (d->boundary_ub) pi.bc_ub[k] = d->boundary_ub[k]; }
    pi.init_state = no_init_t; pi.init_control = no_init_t; pi.init_variable = no_init_v;

    std::lock_guard<std::mutex> lk(g_mu);
    g_ocps.push_back(std::move(o));
    *id = kRuntimeIdBase + (int)g_ocps.size() - 1;
    return CTD_OK;
}

}  // namespace ctd
