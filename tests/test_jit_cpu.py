"""CPU tests of the run-time OCP path (ctd_register_ocp): expression parsing and validation, the generated functor, host
model equality with the registry twin, and that the kernel templates COMPILE for gfx950 with hiprtc (no GPU needed for
that).  Running them is covered by tests/test_gpu_jit.py."""
import numpy as np
import pytest

import ctdirect_jl_amd as ct
import jit_defs


def test_expression_errors_are_reported():
    for bad, frag in ((["x2 + foo(x1)", "x1"], "unknown function"), (["x3", "x1"], "unknown name 'x3'"), (["x1 +", "x1"], "unexpected end"),
                      (["x1^x2", "x1"], "must be a constant"), (["x1^(1+x1)", "x1"], "must be a constant"), (["max(x1)", "x1"], "two arguments"), (["(x1", "x1"], "')' expected"), (["x1 x2", "x1"], "trailing"),
                      (["u1", "x1"], "unknown name 'u1'"), (["xf_1", "x1"], "unknown name")):
        with pytest.raises(ct.CTDirectError) as e:
            ct.register_ocp("bad", dynamics=bad)
        assert e.value.status == ct._lib.CTD_EINVAL and frag in str(e.value), str(e.value)
    with pytest.raises(ct.CTDirectError):
        ct.register_ocp("bad", dynamics=["x1"], nv=1, itf=3)
    with pytest.raises(ct.CTDirectError):
        ct.register_ocp("bad", dynamics=["x1"], constants={"1a": 2.0})


def test_generated_functor_and_traits():
    name = ct.register_ocp("traits", dynamics=["x2*t", "-x1 + k*u1^3/(1 + v1)"], m=1, nv=1, lagrange="2", mayer="xf_1",
                           constants=dict(k=0.5))
    src = ct.ocp_source(name)
    assert "DYN_T = true" in src and "DYN_V = true" in src and "LAG_T = false" in src and "HAS_LAGRANGE = true" in src
    assert "d_powi(u[0], 3)" in src and "return T(2.0);" in src and "0.5" in src and "dx[0] = (x[1] * t);" in src
    # no caller text reaches the compiler except through the parser: identifiers are mapped, numbers re-printed
    with pytest.raises(ct.CTDirectError):
        ct.register_ocp("inject", dynamics=["x1; } evil() {"])


@pytest.mark.parametrize("prob", sorted(jit_defs.TWINS))
def test_host_model_equals_registry_twin(prob):
    """sizes, bounds, patterns (Jacobian: same; Hessian: same, the pattern does not depend on the probe) and the
    default initial guess of a run-time OCP equal those of the compiled registry entry it restates"""
    rt = jit_defs.twin(prob)
    for sch in ("trapeze", "midpoint", "gauss_legendre_2", "gauss_legendre_3_constant_control"):
        a, b = ct.DOCP(prob, 13, sch, device=-1), ct.DOCP(rt, 13, sch, device=-1)
        assert (a.dim_NLP_variables, a.dim_NLP_constraints, a.nnzj, a.nnzh) == (b.dim_NLP_variables, b.dim_NLP_constraints, b.nnzj, b.nnzh)
        for x, y in zip((a.bounds.var_l, a.bounds.var_u, a.bounds.con_l, a.bounds.con_u),
                        (b.bounds.var_l, b.bounds.var_u, b.bounds.con_l, b.bounds.con_u)):
            assert np.array_equal(x, y)
        for f in (ct.DOCP_Jacobian_pattern, ct.DOCP_Hessian_pattern):
            (c1, r1), (c2, r2) = f(a), f(b)
            assert np.array_equal(c1, c2) and np.array_equal(r1, r2)
        assert np.array_equal(ct.initial_guess(a), ct.initial_guess(b))


@pytest.mark.parametrize("sch", ["trapeze", "midpoint", "gauss_legendre_1", "gauss_legendre_2", "gauss_legendre_3_constant_control"])
def test_kernels_compile_for_gfx950_without_a_gpu(sch):
    ct.jit_check(jit_defs.twin("goddard_all"), sch)
    name = ct.register_ocp("vdp_cpu", **jit_defs.VDP) if "vdp_cpu" not in ct.PROBLEMS else "vdp_cpu"
    ct.jit_check(name, sch)


def test_kernels_with_several_controls_per_step_compile_for_gfx950(monkeypatch):
    """control_steps = 3: the midpoint kernels (first order and Hessian) of a run-time OCP with the symbolic stage functions
    (Mayer cost) and of one with a Lagrange cost (second-order numbers, quadrature points with times of their own)"""
    monkeypatch.setenv("CTD_JIT_CHECK_CS", "3")
    ct.jit_check(jit_defs.twin("goddard"), "midpoint")
    ct.jit_check(jit_defs.twin("double_integrator_path"), "midpoint")


def test_compute_without_device_fails_loudly_for_runtime_ocps():
    d = ct.DOCP(jit_defs.twin("goddard"), 10, "midpoint", device=-1)
    with pytest.raises(ct.CTDirectError) as e:
        d.cons(np.full(d.dim_NLP_variables, 0.1))
    assert e.value.status == ct._lib.CTD_ENODEVICE


def test_all_grammar_functions_parse_and_compile():
    """log / tan / atan / tanh / abs are accepted beside exp / sin / cos / sqrt, appear in the generated functor and the kernel
    templates compile for gfx950 with them (first- and second-order number types)"""
    name = "funcs_rt" if "funcs_rt" in ct.PROBLEMS else ct.register_ocp("funcs_rt", **jit_defs.FUNCS)
    src = ct.ocp_source(name)
    for fn in ("d_log(", "d_tan(", "d_atan(", "d_tanh(", "d_abs("):
        assert fn in src, fn
    ct.jit_check(name, "gauss_legendre_2")
    with pytest.raises(ct.CTDirectError) as e:
        ct.register_ocp("bad", dynamics=["asinh(x1)"])
    assert "available: exp, log, sin, cos, tan, atan, tanh, sqrt, abs" in str(e.value)


@pytest.mark.parametrize("expr", [
    "x1*x2*u1 + t*v1^3 - x1/(1 + x2^2)", "exp(-2*x1)*sin(x2 + t) - cos(u1*v1)/(2 + x1^2)",
    "log(1 + x1^2 + u1^2)*tanh(x2 - v1) + atan(x1*x2) + tan(0.3*u1)", "sqrt(1 + x1^2 + x2^2)*abs(u1 - 0.3) + (x1 - x2)^4/(1 + t)",
    "-Cd*x2^2*exp(-beta*(x1 - 1))/x3 - 1/x1^2 + u1*Tmax/x3".replace("Cd", "310").replace("beta", "5").replace("Tmax", "3.5"),
    "((x1 + 2*x2)*(3 - u1))/((1 + v1)*(2 + t)) - -x1", "2^3 + x1*0 + 0*u1 + x2^1 + x1^0",
    # round 4: asin acos sinh cosh floor, real powers, max / min (away from their kinks)
    "asin(0.5*x1*x2) + acos(0.3*u1 - 0.2*x3) + sinh(x1 - v1)*cosh(0.5*x2*t)", "x1^2.5*x2^(-1.5) + (1 + u1^2)^0.5 + x3^-2 + (x1*x2)^(1/3)",
    "max(x1*x2, 0.1*u1)^2 + min(x3 + t, 5 - x1*x1) * x2 + max(0, sin(3*x1))^2", "(x1*8 - floor(x1*8))*x2^2 + floor(x3)*x1*u1",
])
def test_symbolic_second_derivatives_match_finite_differences(expr):
    """the symbolic engine behind the stage functions of run-time OCPs (ctd_sym.hpp): every operator and function of the
    grammar, second derivatives against central differences of the symbolic first derivatives"""
    from emu import emu
    n, m, nv = 3, 1, 1
    rng = np.random.default_rng(4)
    for _ in range(3):
        point = 0.5 + 0.4 * rng.random(1 + n + m + nv)
        err, nnodes = emu.sym_check(expr, n, m, nv, point)
        assert err <= 1e-6 and nnodes < 5000, (expr, err, nnodes)


def test_committed_registry_stage_functions_are_what_the_generator_writes(tmp_path):
    """ctd_sym_registry.hpp is generated (csrc/ctd_gen_sym.cpp: expression form of the registry problems -> symbolic stage
    functions) and committed: rebuilding the tool and running it reproduces the committed file byte for byte"""
    import os
    import subprocess
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ctdirect.jl_amd", "csrc")
    exe = str(tmp_path / "ctd_gen_sym")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-w", "-o", exe, os.path.join(csrc, "ctd_gen_sym.cpp")])
    out = subprocess.run([exe], capture_output=True, text=True, check=True).stdout
    assert out == open(os.path.join(csrc, "ctd_sym_registry.hpp")).read()


def test_problem_name_cannot_inject_source_and_source_is_never_truncated():
    """ADVICE (round 1): the name lands in a comment of the generated source -> only [A-Za-z0-9_.-]; ctd_ocp_source reports the
    size it needs instead of truncating"""
    import ctypes as C
    for bad in ("evil\n} struct X {", "trailing\\", "sp ace", ""):
        with pytest.raises(ct.CTDirectError):
            ct.register_ocp(bad, dynamics=["u1"], n=1, m=1)
    name = ct.register_ocp("ok_name-1.x", dynamics=["u1 - x1"], n=1, m=1)
    src = ct.ocp_source(name)
    assert "struct UserOCP" in src and src.rstrip().endswith("// namespace ctd")      # complete text, not cut
    buf = C.create_string_buffer(16)
    st = ct._lib.lib().ctd_ocp_source(ct.PROBLEMS[name], buf, len(buf))
    assert st == ct._lib.CTD_EINVAL and b"needs" in ct._lib.lib().ctd_last_error(None) and buf.value == b""


def test_host_outputs_are_validated():
    d = ct.DOCP("goddard", 8, "midpoint", device=-1)
    x = np.full(d.dim_NLP_variables, 0.1)
    for bad in (np.zeros(d.dim_NLP_constraints, dtype=np.float32), np.zeros(d.dim_NLP_constraints - 1),
                np.zeros(2 * d.dim_NLP_constraints)[::2]):
        with pytest.raises(ValueError):
            d.cons(x, bad)


def test_committed_assembly_registry_is_what_the_generator_writes(tmp_path):
    """ctd_asm_registry.hpp is generated (csrc/ctd_gen_asm.cpp: the host model's Hessian term tables of the registry problems ->
    straight-line assembly functions of the lane-per-step kernel); the committed header must be what the tool writes from the
    current host model (`make regen_asm` after a change of the term tables)."""
    import os
    import subprocess
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ctdirect.jl_amd", "csrc")
    exe = str(tmp_path / "ctd_gen_asm")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-w", "-I", csrc, "-o", exe, os.path.join(csrc, "ctd_gen_asm.cpp")])
    out = subprocess.check_output([exe]).decode()
    assert out == open(os.path.join(csrc, "ctd_asm_registry.hpp")).read()


def test_round4_grammar_parses_compiles_and_reports_errors():
    """max / min (two arguments), asin acos sinh cosh floor, `^` with any constant exponent, and ALIASES: named sub-expressions in the
    constants string (what the rest of the reference's problem folder needs: bioreactor.jl:15-20, swimmer.jl:39-143)"""
    name = ct.register_ocp("g4_rt", dynamics=["aux*x2 + max(0, sin(w*t))^2", "-x1^1.5 + min(u1, 0.5*x2) + floor(2*t)*0.1 + asin(0.3*x1)"], m=1,
                           lagrange="sinh(0.1*u1)^2 + cosh(x1 - x2) + acos(0.2*aux) + x1^(-0.5) + x2^p", mayer="max(xf_1, xf_2) + xf_1^(1/3)",
                           constants=dict(w=3.0, p=2.5, aux="0.5 + 0.1*inner", inner="cos(x1)*x2"))
    src = ct.ocp_source(name)
    for frag in ("d_max2<T>(", "d_min2<T>(", "d_floor(", "d_asin(", "d_acos(", "d_sinh(", "d_cosh(", "d_powr(", "1.5)", "2.5)", "d_gt("):
        assert frag in src, frag
    assert "DYN_T = true" in src            # (an alias's use of t counts)
    for sch in ("gauss_legendre_2", "midpoint"):
        ct.jit_check(name, sch)
    d = ct.DOCP(name, 7, "trapeze", device=-1, pattern="optimized")
    assert d.nnzj > 0 and d.nnzh > 0
    for kw, frag in ((dict(dynamics=["a*x1"], constants=dict(a="b + 1", b="a*2")), "nested too deeply"),
                     (dict(dynamics=["x1"], constants=dict(x1="2*t")), "collides"),
                     (dict(dynamics=["a"], constants=dict(a="x1 +")), "in alias 'a'"),
                     (dict(dynamics=["x1"], mayer="a", constants=dict(a="x1*2")), "unknown name 'x1'")):
        with pytest.raises(ct.CTDirectError) as e:
            ct.register_ocp("bad", **kw)
        assert frag in str(e.value), str(e.value)


def test_reference_problem_folder_registers_and_builds():
    """every problem of /root/reference/test/problems the registry and round 3's catalogue did not hold: registered from text,
    host model on all schemes and patterns, kernels compile for gfx950 (the large swimmer expressions included)"""
    import problem_folder_defs as pf
    for name in pf.FOLDER:
        rt, _, _ = pf.folder(name)
        for sch in ("trapeze", "midpoint", "gauss_legendre_2", "gauss_legendre_3_constant_control", "euler_implicit"):
            for pattern in ("manual", "structural", "optimized"):
                try:
                    d = ct.DOCP(rt, 9, sch, device=-1, pattern=pattern)
                except ct.CTDirectError:
                    assert pattern == "optimized" and sch == "euler_implicit"
                    continue
                assert d.nnzj > 0 and d.nnzh > 0
    for name, sch in (("swimmer", "gauss_legendre_2"), ("bioreactor_1day", "midpoint"), ("algal_bacterial", "trapeze"), ("action", "gauss_legendre_3"),
                      ("parametric", "midpoint"), ("schlogl", "trapeze")):
        ct.jit_check(pf.folder(name)[0], sch)
