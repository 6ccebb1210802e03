

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
