"""Every entry point of include/mpassit_amd.h called with NULL / zero for every argument: an error code (or the documented no-op of a
destroy / release on NULL), never a crash -- a caller's unset pointer must not take the process, or on this pool the GPU, down.  Runs in
a child process so that a crash is reported with the name of the call that caused it."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import ctypes as C, sys
sys.path.insert(0, %r)
from mpassit_amd import _lib
L = _lib.load()
_lib.init(0)
skip = {"mpg_init", "mpg_finalize", "mpg_last_error", "mpg_warmup_wait", "mpg_comm_idfile_verdict"}   # (the last two: no arguments / returns a string)
zeros = [C.c_void_p(0)] * 14
ok = 0
for name in _lib.SYMBOLS:
    if name in skip:
        continue
    fn = getattr(L, name)
    fn.restype = C.c_int
    print("CALL", name, flush=True)
    rc = fn(*zeros)
    print("RC", name, rc, flush=True)
    ok += 1
# the library is still usable afterwards
buf = C.create_string_buffer(64)
assert L.mpg_device_info(buf, 64, None, None) == 0 and buf.value.startswith(b"gfx")
print("DONE", ok, flush=True)
_lib.finalize()
"""


def test_every_entry_point_survives_null_arguments(gpu_lib):
    r = subprocess.run([sys.executable, "-c", CHILD % ROOT], capture_output=True, text=True, timeout=300)
    calls = [ln.split()[1] for ln in r.stdout.splitlines() if ln.startswith("CALL")]
    done = [ln for ln in r.stdout.splitlines() if ln.startswith("DONE")]
    assert r.returncode == 0 and done, "crashed in %s (rc %d)\n%s" % (calls[-1] if calls else "?", r.returncode, r.stderr[-2000:])
    rcs = {ln.split()[1]: int(ln.split()[2]) for ln in r.stdout.splitlines() if ln.startswith("RC")}
    assert len(rcs) >= 70
    # NULL objects: destroying / releasing them is a no-op, everything else is refused
    noop = {n for n in rcs if n.endswith("_destroy") or n.endswith("_release") or n in ("mpg_dev_free",)}
    wrong = {n: rc for n, rc in rcs.items() if (rc == 0) != (n in noop) and n not in ("mpg_tune", "mpg_comm_virtual_stats")}
    accepted = {n: rc for n, rc in wrong.items() if rc == 0}
    assert not accepted or set(accepted) <= {"mpg_bswap_dev", "mpg_debug_scan_i32", "mpg_pack_dev", "mpg_pack_rows_dev", "mpg_handle_cache_clear",
                                           "mpg_dev_download", "mpg_dev_upload", "mpg_post_cast_dev", "mpg_device_info"}, accepted   # zero elements / every pointer optional: a no-op
