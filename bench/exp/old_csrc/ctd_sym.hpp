// ctd_sym.hpp -- small symbolic engine for run-time OCPs (host only).
//
// The expressions of a run-time OCP (ctd_jit.cpp) are kept as a hash-consed DAG.  For the Hessian of the Lagrangian the
// scalar a stage-type evaluation point contributes,
//     Phi = sum_r W_r f_r(t, x, u, v) + (cost weight) * l(t, x, u, v)              (times the step length where the scheme says so)
// with the evaluation point written as a function of the differentiation variables of the kernel's records (state, control
// and optimisation-variable directions; the latter also move the time grid), is differentiated twice SYMBOLICALLY and the
// structurally nonzero second derivatives are emitted as straight-line code with shared sub-expressions: one lane then
// produces every second derivative of a point in one pass instead of one second-order forward-number evaluation per pair
// of directions (ctd_hess_body.hpp).  In the reference these derivatives come from ADNLPModels' sparse Hessian backend over
// the closures (src/collocation.jl:121-125); the values are the same derivatives, so parity is unchanged.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <map>
#include <string>
#include <tuple>
#include <vector>

namespace ctd {
namespace sym {

// POWR: a ^ (constant real exponent; b = the exponent's CONST node).  MAX / MIN / GT (1.0 where a > b, else 0.0): binary; the
// derivative of max / min selects with GT (ForwardDiff's convention, d_gt in ctd_common.hpp)
enum Op : uint8_t { CONST, PARAM, VAR, ADD, SUB, MUL, DIV, NEG, POWI, FN, POWR, MAX, MIN, GT };
enum Fn : uint8_t { F_EXP, F_LOG, F_SIN, F_COS, F_TAN, F_ATAN, F_TANH, F_SQRT, F_ABS, F_SGN, F_ASIN, F_ACOS, F_SINH, F_COSH, F_FLOOR };

struct Node { Op op; int a, b; double c; };   // PARAM / VAR: a = index; POWI: b = exponent; FN: b = Fn id; POWR: b = exponent node

class Graph {
public:
    std::vector<Node> nodes;
    int constant(double v) {
        if (v == 0.0) v = 0.0;                                   // -0.0 and 0.0 are one node
        auto it = consts_.find(v);
        if (it != consts_.end()) return it->second;
        const int id = push(Node{CONST, -1, -1, v});
        consts_[v] = id;
        return id;
    }
    int param(int i) { return intern(PARAM, i, -1); }
    int var(int i) { return intern(VAR, i, -1); }
    bool is_const(int n) const { return nodes[n].op == CONST; }
    double cval(int n) const { return nodes[n].c; }
    bool is_zero(int n) const { return is_const(n) && cval(n) == 0.0; }
    bool is_one(int n) const { return is_const(n) && cval(n) == 1.0; }

    int add(int a, int b) {
        if (is_zero(a)) return b;
        if (is_zero(b)) return a;
        if (is_const(a) && is_const(b)) return constant(cval(a) + cval(b));
        if (nodes[b].op == NEG) return sub(a, nodes[b].a);
        if (a > b) std::swap(a, b);                              // commutative: one node for a + b and b + a
        return intern(ADD, a, b);
    }
    int sub(int a, int b) {
        if (is_zero(b)) return a;
        if (is_zero(a)) return neg(b);
        if (a == b) return constant(0.0);
        if (is_const(a) && is_const(b)) return constant(cval(a) - cval(b));
        if (nodes[b].op == NEG) return add(a, nodes[b].a);
        return intern(SUB, a, b);
    }
    int neg(int a) {
        if (is_const(a)) return constant(-cval(a));
        if (nodes[a].op == NEG) return nodes[a].a;
        return intern(NEG, a, -1);
    }
    int mul(int a, int b) {
        if (is_zero(a) || is_zero(b)) return constant(0.0);
        if (is_one(a)) return b;
        if (is_one(b)) return a;
        if (is_const(a) && is_const(b)) return constant(cval(a) * cval(b));
        if (is_const(a) && cval(a) == -1.0) return neg(b);
        if (is_const(b) && cval(b) == -1.0) return neg(a);
        if (nodes[a].op == NEG && nodes[b].op == NEG) return mul(nodes[a].a, nodes[b].a);
        if (nodes[a].op == NEG) return neg(mul(nodes[a].a, b));
        if (nodes[b].op == NEG) return neg(mul(a, nodes[b].a));
        if (a > b) std::swap(a, b);
        return intern(MUL, a, b);
    }
    // a / b as a * (1 / b): one division per distinct denominator, shared by every quotient that uses it
    int div(int a, int b) {
        if (is_zero(a)) return constant(0.0);
        if (is_one(b)) return a;
        if (is_const(a) && is_const(b)) return constant(cval(a) / cval(b));
        if (is_const(b)) return mul(a, constant(1.0 / cval(b)));
        const int one = constant(1.0);
        const int inv = intern(DIV, one, b);
        return a == one ? inv : mul(a, inv);
    }
    int powi(int a, int k) {
        if (k == 0) return constant(1.0);
        if (k == 1) return a;
        if (is_const(a)) { double r = 1.0; for (int i = 0; i < k; ++i) r *= cval(a); return constant(r); }
        return intern(POWI, a, k);
    }
    int powr(int a, double p) {
        if (p == 0.0) return constant(1.0);
        if (p == 1.0) return a;
        if (p == std::floor(p) && p >= 2.0 && p <= 64.0) return powi(a, (int)p);
        if (is_const(a)) return constant(std::pow(cval(a), p));
        return intern(POWR, a, constant(p));
    }
    int gt(int a, int b) {
        if (is_const(a) && is_const(b)) return constant(cval(a) > cval(b) ? 1.0 : 0.0);
        return intern(GT, a, b);
    }
    int max2(int a, int b) {
        if (is_const(a) && is_const(b)) return constant(cval(a) > cval(b) ? cval(a) : cval(b));
        return intern(MAX, a, b);
    }
    int min2(int a, int b) {
        if (is_const(a) && is_const(b)) return constant(cval(a) > cval(b) ? cval(b) : cval(a));
        return intern(MIN, a, b);
    }
    int fn(Fn f, int a) {
        if (is_const(a)) {
            const double x = cval(a);
            switch (f) {
                case F_ASIN: return constant(std::asin(x));
                case F_ACOS: return constant(std::acos(x));
                case F_SINH: return constant(std::sinh(x));
                case F_COSH: return constant(std::cosh(x));
                case F_FLOOR: return constant(std::floor(x));
                case F_EXP: return constant(std::exp(x));
                case F_LOG: return constant(std::log(x));
                case F_SIN: return constant(std::sin(x));
                case F_COS: return constant(std::cos(x));
                case F_TAN: return constant(std::tan(x));
                case F_ATAN: return constant(std::atan(x));
                case F_TANH: return constant(std::tanh(x));
                case F_SQRT: return constant(std::sqrt(x));
                case F_ABS: return constant(std::fabs(x));
                case F_SGN: return constant(x > 0.0 ? 1.0 : (x < 0.0 ? -1.0 : 0.0));
            }
        }
        return intern(FN, a, (int)f);
    }

    // d node / d VAR(v)
    int diff(int n, int v) {
        const auto key = std::make_pair(n, v);
        auto it = dmemo_.find(key);
        if (it != dmemo_.end()) return it->second;
        const Node nd = nodes[n];
        int r = constant(0.0);
        switch (nd.op) {
            case CONST: case PARAM: break;
            case VAR: r = constant(nd.a == v ? 1.0 : 0.0); break;
            case ADD: r = add(diff(nd.a, v), diff(nd.b, v)); break;
            case SUB: r = sub(diff(nd.a, v), diff(nd.b, v)); break;
            case NEG: r = neg(diff(nd.a, v)); break;
            case MUL: r = add(mul(diff(nd.a, v), nd.b), mul(nd.a, diff(nd.b, v))); break;
            case DIV: {                                          // (a / b)' = (a' - (a / b) b') / b
                const int da = diff(nd.a, v), db = diff(nd.b, v);
                r = div(sub(da, mul(n, db)), nd.b);
                break;
            }
            case POWI: r = mul(mul(constant((double)nd.b), powi(nd.a, nd.b - 1)), diff(nd.a, v)); break;
            case POWR: { const double p = cval(nd.b); r = mul(mul(constant(p), powr(nd.a, p - 1.0)), diff(nd.a, v)); break; }
            case GT: break;
            case MAX: case MIN: {            // max: a > b ? a' : b';  min: a > b ? b' : a'
                const int da = diff(nd.a, v), db = diff(nd.b, v);
                if (is_zero(da) && is_zero(db)) break;
                const int g1 = gt(nd.a, nd.b), g0 = sub(constant(1.0), g1);
                r = nd.op == MAX ? add(mul(g1, da), mul(g0, db)) : add(mul(g0, da), mul(g1, db));
                break;
            }
            case FN: {
                const int da = diff(nd.a, v);
                if (is_zero(da)) break;
                int f1 = constant(0.0);
                switch ((Fn)nd.b) {
                    case F_EXP: f1 = n; break;
                    case F_LOG: f1 = div(constant(1.0), nd.a); break;
                    case F_SIN: f1 = fn(F_COS, nd.a); break;
                    case F_COS: f1 = neg(fn(F_SIN, nd.a)); break;
                    case F_TAN: f1 = add(constant(1.0), mul(n, n)); break;
                    case F_ATAN: f1 = div(constant(1.0), add(constant(1.0), mul(nd.a, nd.a))); break;
                    case F_TANH: f1 = sub(constant(1.0), mul(n, n)); break;
                    case F_SQRT: f1 = div(constant(0.5), n); break;
                    case F_ABS: f1 = fn(F_SGN, nd.a); break;
                    case F_SGN: case F_FLOOR: break;
                    case F_ASIN: f1 = div(constant(1.0), fn(F_SQRT, sub(constant(1.0), mul(nd.a, nd.a)))); break;
                    case F_ACOS: f1 = neg(div(constant(1.0), fn(F_SQRT, sub(constant(1.0), mul(nd.a, nd.a))))); break;
                    case F_SINH: f1 = fn(F_COSH, nd.a); break;
                    case F_COSH: f1 = fn(F_SINH, nd.a); break;
                }
                r = mul(f1, da);
                break;
            }
        }
        dmemo_[key] = r;
        return r;
    }

    // the node with every VAR replaced by the constant 0 (the derivatives are taken at the evaluation point itself)
    int at_zero(int n) {
        auto it = zmemo_.find(n);
        if (it != zmemo_.end()) return it->second;
        const Node nd = nodes[n];
        int r = n;
        switch (nd.op) {
            case CONST: case PARAM: break;
            case VAR: r = constant(0.0); break;
            case ADD: r = add(at_zero(nd.a), at_zero(nd.b)); break;
            case SUB: r = sub(at_zero(nd.a), at_zero(nd.b)); break;
            case NEG: r = neg(at_zero(nd.a)); break;
            case MUL: r = mul(at_zero(nd.a), at_zero(nd.b)); break;
            case DIV: r = div(at_zero(nd.a), at_zero(nd.b)); break;
            case POWI: r = powi(at_zero(nd.a), nd.b); break;
            case POWR: r = powr(at_zero(nd.a), cval(nd.b)); break;
            case MAX: r = max2(at_zero(nd.a), at_zero(nd.b)); break;
            case MIN: r = min2(at_zero(nd.a), at_zero(nd.b)); break;
            case GT: r = gt(at_zero(nd.a), at_zero(nd.b)); break;
            case FN: r = fn((Fn)nd.b, at_zero(nd.a)); break;
        }
        zmemo_[n] = r;
        return r;
    }

    double eval(int n, const std::vector<double>& prm, const std::vector<double>& vars) const {
        std::map<int, double> memo;
        return eval_rec(n, prm, vars, memo);
    }

private:
    double eval_rec(int n, const std::vector<double>& prm, const std::vector<double>& vars, std::map<int, double>& memo) const {
        auto it = memo.find(n);
        if (it != memo.end()) return it->second;
        const double r = eval_node(n, prm, vars, memo);
        memo[n] = r;
        return r;
    }
    double eval_node(int n, const std::vector<double>& prm, const std::vector<double>& vars, std::map<int, double>& memo) const {
        auto eval = [&](int c, const std::vector<double>&, const std::vector<double>&) { return eval_rec(c, prm, vars, memo); };
        const Node& nd = nodes[n];
        switch (nd.op) {
            case CONST: return nd.c;
            case PARAM: return prm[nd.a];
            case VAR: return vars[nd.a];
            case ADD: return eval(nd.a, prm, vars) + eval(nd.b, prm, vars);
            case SUB: return eval(nd.a, prm, vars) - eval(nd.b, prm, vars);
            case NEG: return -eval(nd.a, prm, vars);
            case MUL: return eval(nd.a, prm, vars) * eval(nd.b, prm, vars);
            case DIV: return eval(nd.a, prm, vars) / eval(nd.b, prm, vars);
            case POWI: { const double x = eval(nd.a, prm, vars); double r = 1.0; for (int i = 0; i < nd.b; ++i) r *= x; return r; }
            case POWR: return std::pow(eval(nd.a, prm, vars), nodes[nd.b].c);
            case MAX: { const double x = eval(nd.a, prm, vars), y = eval(nd.b, prm, vars); return x > y ? x : y; }
            case MIN: { const double x = eval(nd.a, prm, vars), y = eval(nd.b, prm, vars); return x > y ? y : x; }
            case GT: return eval(nd.a, prm, vars) > eval(nd.b, prm, vars) ? 1.0 : 0.0;
            case FN: {
                const double x = eval(nd.a, prm, vars);
                switch ((Fn)nd.b) {
                    case F_ASIN: return std::asin(x);
                    case F_ACOS: return std::acos(x);
                    case F_SINH: return std::sinh(x);
                    case F_COSH: return std::cosh(x);
                    case F_FLOOR: return std::floor(x);
                    case F_EXP: return std::exp(x);
                    case F_LOG: return std::log(x);
                    case F_SIN: return std::sin(x);
                    case F_COS: return std::cos(x);
                    case F_TAN: return std::tan(x);
                    case F_ATAN: return std::atan(x);
                    case F_TANH: return std::tanh(x);
                    case F_SQRT: return std::sqrt(x);
                    case F_ABS: return std::fabs(x);
                    case F_SGN: return x > 0.0 ? 1.0 : (x < 0.0 ? -1.0 : 0.0);
                }
            }
        }
        return 0.0;
    }

public:
    // straight-line C++ for the listed outputs: "lhs = expr;" lines preceded by one temporary per shared interior node.
    // `prm` is the name of the parameter array.  No VAR may be reachable (take at_zero first).
    std::string codegen(const std::vector<std::pair<std::string, int>>& outputs, const std::string& prm, const std::string& indent) const {
        std::vector<int> uses(nodes.size(), 0);
        std::vector<char> seen(nodes.size(), 0);
        std::vector<int> order;
        for (auto& o : outputs) visit(o.second, uses, seen, order);
        for (auto& o : outputs) ++uses[o.second];
        std::vector<std::string> name(nodes.size());
        std::string s;
        int ntmp = 0;
        // sin and cos of the same argument (every rotation): one d_sincos call for the pair -- the two library calls do not share
        // their range reduction once inlined (measured: six reductions per part of the 12-state quadrotor's dynamics instead of three)
        std::map<int, std::pair<int, int>> trig;                      // argument node -> (sin node, cos node), -1 = absent
        for (int n : order)
            if (nodes[n].op == FN && (nodes[n].b == 2 || nodes[n].b == 3)) {
                auto& pr = trig.emplace(nodes[n].a, std::make_pair(-1, -1)).first->second;
                (nodes[n].b == 2 ? pr.first : pr.second) = n;
            }
        for (int n : order) {
            const Node& nd = nodes[n];
            if (!name[n].empty()) continue;                            // (named together with its sin / cos partner)
            if (nd.op == FN && (nd.b == 2 || nd.b == 3)) {
                const auto pr = trig[nd.a];
                if (pr.first >= 0 && pr.second >= 0) {
                    name[pr.first] = "s" + std::to_string(ntmp++);
                    name[pr.second] = "s" + std::to_string(ntmp++);
                    s += indent + "double " + name[pr.first] + ", " + name[pr.second] + "; d_sincos(" + name[nd.a] + ", " + name[pr.first] + ", " +
                         name[pr.second] + ");\n";
                    continue;
                }
            }
            if (nd.op == CONST) { name[n] = num(nd.c); continue; }
            if (nd.op == PARAM) { name[n] = prm + "[" + std::to_string(nd.a) + "]"; continue; }
            std::string e;
            switch (nd.op) {
                case ADD: e = name[nd.a] + " + " + name[nd.b]; break;
                case SUB: e = name[nd.a] + " - " + name[nd.b]; break;
                case MUL: e = name[nd.a] + " * " + name[nd.b]; break;
                case DIV: e = name[nd.a] + " / " + name[nd.b]; break;
                case NEG: e = "-" + name[nd.a]; break;
                case POWI: e = "d_powi(" + name[nd.a] + ", " + std::to_string(nd.b) + ")"; break;
                case POWR: e = "d_powr(" + name[nd.a] + ", " + name[nd.b] + ")"; break;
                case MAX: e = "d_max(" + name[nd.a] + ", " + name[nd.b] + ")"; break;
                case MIN: e = "d_min(" + name[nd.a] + ", " + name[nd.b] + ")"; break;
                case GT: e = "d_gt(" + name[nd.a] + ", " + name[nd.b] + ")"; break;
                case FN: {
                    static const char* fnn[] = {"d_exp", "d_log", "d_sin", "d_cos", "d_tan", "d_atan", "d_tanh", "d_sqrt", "d_abs", "d_sgn",
                                                "d_asin", "d_acos", "d_sinh", "d_cosh", "d_floor"};
                    e = std::string(fnn[nd.b]) + "(" + name[nd.a] + ")";
                    break;
                }
                default: e = "0.0"; break;
            }
            if (uses[n] > 1 || nd.op == FN || nd.op == DIV || nd.op == POWI || nd.op == POWR || nd.op == MAX || nd.op == MIN || nd.op == GT) {
                name[n] = "s" + std::to_string(ntmp++);
                s += indent + "const double " + name[n] + " = " + e + ";\n";
            } else {
                name[n] = "(" + e + ")";
            }
        }
        for (auto& o : outputs) s += indent + o.first + " = " + name[o.second] + ";\n";
        return s;
    }

    static std::string num(double v) {
        char buf[64];
        std::snprintf(buf, sizeof buf, "%.17g", v);
        std::string r(buf);
        if (r.find_first_of(".eEn") == std::string::npos) r += ".0";
        if (v < 0) r = "(" + r + ")";
        return r;
    }

private:
    std::map<std::tuple<int, int, int>, int> table_;
    std::map<double, int> consts_;
    std::map<std::pair<int, int>, int> dmemo_;
    std::map<int, int> zmemo_;
    int push(const Node& n) { nodes.push_back(n); return (int)nodes.size() - 1; }
    int intern(Op op, int a, int b) {
        const auto key = std::make_tuple((int)op, a, b);
        auto it = table_.find(key);
        if (it != table_.end()) return it->second;
        const int id = push(Node{op, a, b, 0.0});
        table_[key] = id;
        return id;
    }
    void visit(int n, std::vector<int>& uses, std::vector<char>& seen, std::vector<int>& order) const {
        if (seen[n]) return;
        seen[n] = 1;
        const Node& nd = nodes[n];
        if (nd.op >= ADD) {
            visit(nd.a, uses, seen, order); ++uses[nd.a];
            if (nd.op == ADD || nd.op == SUB || nd.op == MUL || nd.op == DIV || nd.op == POWR || nd.op == MAX || nd.op == MIN || nd.op == GT) { visit(nd.b, uses, seen, order); ++uses[nd.b]; }
        }
        order.push_back(n);
    }
};

}  // namespace sym
}  // namespace ctd
