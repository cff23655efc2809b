"""tests/spawner.py (the helper that starts the GPU tests' child processes before the test process touches the GPU) on the CPU."""
import sys

import spawner


def test_the_helper_runs_commands_and_reports_failures():
    spawner.stop()
    # without a helper the command is started from this process
    r = spawner.run([sys.executable, "-c", "print('direct')"])
    assert r.returncode == 0 and r.stdout.strip() == "direct"
    spawner.start()
    try:
        r = spawner.run([sys.executable, "-c", "import os, sys; print(os.environ['SPAWNER_PROBE']); sys.stderr.write('e'); sys.exit(3)"],
                        env={"SPAWNER_PROBE": "x", "PATH": "/usr/bin:/bin"})
        assert (r.returncode, r.stdout.strip(), r.stderr) == (3, "x", "e")
        r = spawner.run([sys.executable, "-c", "import time; time.sleep(5)"], timeout=0.5)
        assert r.returncode == -999 and "timeout" in r.stderr
        r = spawner.run(["/nonexistent/program"])
        assert r.returncode == -998
        # the helper survives all of that, and `check` raises like subprocess.run does
        assert spawner.run([sys.executable, "-c", "print(1)"]).stdout.strip() == "1"
        import subprocess
        try:
            spawner.run([sys.executable, "-c", "raise SystemExit(2)"], check=True)
            assert False, "check=True must raise"
        except subprocess.CalledProcessError as e:
            assert e.returncode == 2
    finally:
        spawner.stop()
