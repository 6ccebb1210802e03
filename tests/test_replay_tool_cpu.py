"""tools/replay_reference_dump.py is the route by which "parity unpinned" gets closed on a box with cargo; nobody can run the Rust
side here, so the tool is at least proven to work: fed a dump synthesised from the oracle it must pass every check, name the Merkle
node rule the dump was made with, and reject a corrupted known answer."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_replay_tool_self_test(oracle):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "replay_reference_dump.py"), "--self-test"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "SELF-TEST OK" in r.stdout and r.stdout.count("hash_mode=[0]") == 2 and "hash_mode=[1]" in r.stdout   # mode 0 twice: the second run is the corrupted dump
    assert "FAIL interpolate (oracle)" in r.stdout
