import os, subprocess, sys, faulthandler
faulthandler.enable()
print("LD_PRELOAD=", os.environ.get("LD_PRELOAD"), flush=True)
mode = sys.argv[1]
if mode != "nogpu":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from softwarerenderer_amd import Device
    d = Device(0)
    print("gpu initialised", flush=True)
if mode == "spawn":
    pid = os.posix_spawn("/bin/echo", ["/bin/echo", "child via posix_spawn ok"], os.environ)
    print("waitpid", os.waitpid(pid, 0), flush=True)
else:
    r = subprocess.run(["/bin/echo", "child via subprocess ok"], capture_output=True, text=True)
    print(r.returncode, r.stdout, r.stderr, flush=True)
print("parent alive", mode, flush=True)
