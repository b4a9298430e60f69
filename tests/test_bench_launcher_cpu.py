"""bench.py's own launcher (`python bench.py --gpus N` without torch.distributed.run): it must watch ALL ranks -- a rank that
dies while rank 0 waits in a collective ends the run within seconds with that rank's exit code -- and
it must not touch the GPU runtime itself (no torch import in the launching process)."""
import os
import subprocess
import sys
import textwrap
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(tmp_path, body, gpus=3, env=None):
    script = tmp_path / "rank.py"
    script.write_text(textwrap.dedent(body))
    code = (f"import sys; sys.path.insert(0, {ROOT!r}); import bench; "
            f"rc = bench.self_launch([], {gpus}, script={str(script)!r}); "
            "assert 'torch' not in sys.modules, 'the launcher imported torch'; sys.exit(rc)")
    t0 = time.monotonic()
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, env=dict(os.environ, **(env or {})))
    return p, time.monotonic() - t0


def test_a_dead_rank_ends_the_run_within_seconds(tmp_path):
    p, el = _run(tmp_path, """
        import os, sys, time
        r = int(os.environ["RANK"])
        if r == 1:
            time.sleep(0.5)
            os._exit(17)
        time.sleep(600)          # ranks 0 and 2 "wait in a collective"
    """)
    assert p.returncode == 17, (p.returncode, p.stderr)
    assert el < 10.0, el
    assert "rank 1 exited with code 17" in p.stderr


def test_all_ranks_fine_relays_rank0_line_only(tmp_path):
    p, el = _run(tmp_path, """
        import os
        print('{"rank": %s, "world": %s, "addr": "%s"}' % (os.environ["RANK"], os.environ["WORLD_SIZE"], os.environ["MASTER_ADDR"]))
    """)
    assert p.returncode == 0, p.stderr
    assert p.stdout.strip() == '{"rank": 0, "world": 3, "addr": "127.0.0.1"}'


def test_deadline_ends_hung_ranks(tmp_path):
    p, el = _run(tmp_path, "import time; time.sleep(600)", gpus=2, env={"CTD_BENCH_DEADLINE_S": "1.5"})
    assert p.returncode == 124 and el < 15.0, (p.returncode, el)


def test_rank_killed_by_a_signal_is_reported(tmp_path):
    p, el = _run(tmp_path, """
        import os, signal, time
        if os.environ["RANK"] == "0":
            os.kill(os.getpid(), signal.SIGKILL)
        time.sleep(600)
    """, gpus=2)
    assert p.returncode == 128 + 9 and el < 10.0, (p.returncode, el)
