"""Out-of-bounds hunt on guard pages: tests/guard/run_cases.py runs whole optimizer steps of the small
models in every arithmetic mode plus single layers on odd shapes in a process whose EVERY device allocation has unmapped pages on both
sides (tests/guard/guard_alloc.cpp, hooked into torch through CUDAPluggableAllocator).  A kernel that reads or writes one 16-byte piece
past the end ('hi' placement) or before the start ('lo') of any operand kills that process with `Memory access fault by GPU`; the
parent names the case, restarts behind it and fails with the list."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RUN = os.path.join(ROOT, "tests", "guard", "run_cases.py")


def sweep(mode, first=0, last=None, fill=None, timeout=1500):
    """-> (faulting cases, python-level errors, number of cases): restarts behind every case that killed the process."""
    env = dict(os.environ, AMD_SERIALIZE_KERNEL="3", PYTHONFAULTHANDLER="1")
    env.pop("DPIG_GUARD", None)
    if fill is not None:
        env["DPIG_GUARD_FILL"] = str(fill)
    faults, errs, total = [], [], None
    while True:
        cmd = [sys.executable, RUN, mode, str(first)] + ([str(last)] if last is not None else [])
        r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
        cur = None
        for l in r.stdout.splitlines():
            if l.startswith("CASE "):
                cur = l.split(" ", 2)
            elif l.startswith("OK "):
                cur = None
            elif l.startswith("ERR "):
                errs.append(l)
                cur = None
            elif l.startswith("DONE "):
                total = int(l.split()[1])
        if total is not None and r.returncode == 0:
            return faults, errs, total
        if cur is None:        # died outside a case (start-up): nothing to restart behind
            raise AssertionError("guard run died outside a case (rc %s):\n%s\n%s" % (r.returncode, r.stdout[-1500:], r.stderr[-3000:]))
        faults.append("%s %s [rc %s] %s" % (cur[1], cur[2], r.returncode,
                                            " | ".join(l for l in r.stderr.splitlines() if "fault" in l.lower() or "File \"" in l)[-1200:]))
        first = int(cur[1]) + 1
        if last is not None and first > last:
            return faults, errs, total


@pytest.mark.parametrize("mode", ["hi", "lo"])
def test_no_kernel_touches_memory_outside_its_operands(dev, mode):
    faults, errs, total = sweep(mode)
    assert not faults, "out-of-bounds access in %d case(s) [%s placement]:\n%s" % (len(faults), mode, "\n".join(faults))
    assert not errs, "\n".join(errs)
    assert total and total > 200


def test_guard_allocator_catches_a_one_piece_overrun(dev):
    """The harness itself: a torch kernel reading 16 bytes past a guarded tensor faults; the same read inside the tensor does not."""
    code = r'''
import sys, os
sys.path.insert(0, os.path.join(%r, "tests"))
import conftest
conftest.install_guard_allocator("hi")
import torch, ctypes
t = torch.zeros(1024, dtype=torch.float32, device="cuda:0")
torch.cuda.synchronize()
hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
host = ctypes.create_string_buffer(16)
off = int(sys.argv[1])
hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
rc = hip.hipMemcpy(host, ctypes.c_void_p(t.data_ptr() + off), 16, 2)
print("RC", rc, flush=True)
sys.exit(0 if rc == 0 else 3)
''' % ROOT
    inside = subprocess.run([sys.executable, "-c", code, str(4096 - 16)], capture_output=True, text=True, timeout=300)
    assert inside.returncode == 0 and "RC 0" in inside.stdout, inside.stdout + inside.stderr[-1500:]
    beyond = subprocess.run([sys.executable, "-c", code, str(4096)], capture_output=True, text=True, timeout=300)
    assert beyond.returncode != 0, beyond.stdout + beyond.stderr[-1500:]
