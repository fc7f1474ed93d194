"""bench.py's loop schedule and its self-launching entry (CPU; nothing here touches a GPU).

The timed region of the driver's command line (``--steps 20 --warmup 5``, callback_freq 10) must not be
the first execution of any code path of a LinSys iteration - round 3 met its first logging iteration
inside the timed region and charged 81 ms of one-time loads to the metric."""

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import bench  # noqa: E402


def _trace(warmup, steps, freq):
    """Run the schedule with recording fakes.  Paths of an iteration: a plain step, a logging pass, and the
    step that follows a logging pass (it consumes the product computed ahead by the metric)."""
    events, state = [], {"after_log": False, "timed": False}

    def step():
        events.append(("step_after_log" if state["after_log"] else "step_plain", state["timed"]))
        state["after_log"] = False

    def log():
        events.append(("log", state["timed"]))
        state["after_log"] = True

    def start():
        state["timed"] = True

    def stop():
        state["timed"] = False
        return 1.0

    assert bench.run_schedule(step, log, warmup, steps, freq, start, stop) == 1.0
    return events


@pytest.mark.parametrize("warmup,freq", [(5, 10), (3, 10), (0, 10), (10, 10), (0, 1), (7, 3)])
def test_every_path_of_an_iteration_runs_before_the_timed_region(warmup, freq):
    ev = _trace(warmup, 20, freq)
    untimed = {name for name, timed in ev if not timed}
    assert {"step_plain", "log", "step_after_log"} <= untimed
    # each at least twice (the first execution loads code objects, the second sizes caches / workspaces)
    for name in ("step_plain", "log", "step_after_log"):
        assert sum(1 for nm, timed in ev if nm == name and not timed) >= 2, name


@pytest.mark.parametrize("warmup,steps,freq", [(5, 20, 10), (0, 7, 10), (3, 50, 10), (10, 50, 10)])
def test_timed_region_is_exactly_k_steps_with_the_reference_logging_cadence(warmup, steps, freq):
    ev = _trace(warmup, steps, freq)
    timed = [name for name, t in ev if t]
    assert sum(1 for nm in timed if nm.startswith("step")) == steps
    expect_logs = sum(1 for i in range(warmup + 1, warmup + steps + 1) if i % freq == 0)
    assert timed.count("log") == expect_logs


def test_gpus_n_without_a_launcher_starts_children_and_never_execs(monkeypatch):
    """``python bench.py --gpus 2`` with no WORLD_SIZE: torch.distributed.run is started as a CHILD
    (subprocess), one rank per GPU on 127.0.0.1, the script's own flags passed through, --n moved to the
    environment (the launcher's parser claims --n* prefixes); the exit code is the child's."""
    import subprocess

    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7

    monkeypatch.setattr(subprocess, "call", fake_call)
    for name in ("execv", "execve", "execvp", "execvpe", "execl", "execle", "execlp"):
        monkeypatch.setattr(os, name, lambda *a, **k: (_ for _ in ()).throw(AssertionError("exec is forbidden")))
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "20", "--warmup", "5", "--n", "60000"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7
    cmd = seen["cmd"]
    assert cmd[0] == sys.executable and cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nproc-per-node=2" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    script = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[script + 1:] == ["--gpus", "2", "--steps", "20", "--warmup", "5"]
    assert seen["env"]["RK_BENCH_N"] == "60000"
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_world_size_mismatch_is_an_error(monkeypatch):
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert "WORLD_SIZE=2" in str(e.value.code)
