

def test_guardian_prints_the_line_when_the_process_dies_and_stays_silent_otherwise():
    """bench.py's one-proof attempt drives RCCL paths that no 1-GPU box has run; if rank 0 dies in it the finished headline line must still
    come out (bench.py _arm_guardian): a child that waits on a pipe prints the fallback when the parent is gone, nothing when disarmed."""
    import subprocess, sys, os, json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    died = subprocess.run([sys.executable, "-c", "import os, signal, bench; bench._arm_guardian({'value': 1, 'one_proof': {'error': 'died'}}); os.kill(os.getpid(), signal.SIGKILL)"],
                          cwd=root, capture_output=True, text=True, timeout=60)
    lines = [l for l in died.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and json.loads(lines[0]) == {"value": 1, "one_proof": {"error": "died"}}, (died.stdout, died.stderr)
    fine = subprocess.run([sys.executable, "-c", "import json, bench; g = bench._arm_guardian({'value': 1, 'one_proof': {'error': 'died'}}); bench._disarm_guardian(g); print(json.dumps({'value': 1, 'one_proof': {'ok': True}}))"],
                          cwd=root, capture_output=True, text=True, timeout=60)
    lines = [l for l in fine.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and json.loads(lines[0])["one_proof"] == {"ok": True}, (fine.stdout, fine.stderr)


def test_plain_start_with_several_gpus_builds_the_launch_command(monkeypatch):
    """bench.py --gpus N started without a launcher hands the same arguments to torch.distributed.run (one process per GPU, 127.0.0.1,
    a free port) and returns its exit code; with WORLD_SIZE set (it IS a rank) nothing is launched.  (The launch itself runs in the
    GPU suite: test_bench_started_plainly_with_gpus_2_launches_its_own_ranks.)"""
    import subprocess, sys, os
    import bench
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7
    monkeypatch.setattr(subprocess, "call", fake_call)
    assert bench.self_launch(8, ["--gpus", "8", "--steps", "2"]) == 7
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    assert cmd[-5:] == [os.path.abspath(bench.__file__), "--gpus", "8", "--steps", "2"]
    assert seen["env"]["NX_BENCH_LAUNCHER"].startswith("self") and seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    # main(): --gpus 2 without WORLD_SIZE -> self_launch; its code is the process's
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "1"])
    try:
        bench.main()
        assert False, "main() should have exited through the launcher"
    except SystemExit as e:
        assert e.code == 7 and seen["cmd"][-4:] == ["--gpus", "2", "--steps", "1"]
