"""-np N rendezvous of the host driver (host/idfile.c): rank 0 hands the 256-byte communicator id to the other
ranks through a file.  A file left behind by a run that died must never be taken for this launch's id (the ranks would
call ncclCommInitRank with different ids and hang).  CPU only: the protocol is plain file I/O."""
import ctypes as C
import os
import subprocess
import sys
import time

from conftest import ROOT

HOST = os.path.join(ROOT, "host")

WAITER = r"""
import ctypes as C, os, sys
L = C.CDLL(sys.argv[1])
L.mk_idfile_wait.argtypes = [C.c_char_p, C.c_char_p, C.c_double, C.c_double]
buf = C.create_string_buffer(256)
rc = L.mk_idfile_wait(sys.argv[2].encode(), buf, float(sys.argv[3]), float(sys.argv[4]))
sys.stdout.write("%d %s" % (rc, buf.raw.hex()))
"""


def _lib():
    so = os.path.join(HOST, "libmkhost.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", HOST], stdout=subprocess.DEVNULL)
    L = C.CDLL(so)
    L.mk_idfile_publish.argtypes = [C.c_char_p, C.c_char_p]
    L.mk_idfile_clear.argtypes = [C.c_char_p]
    return L, so


def _env(**kw):
    env = {k: v for k, v in os.environ.items()
           if k not in ("MK_LAUNCH_ID", "SLURM_JOB_ID", "SLURM_STEP_ID", "PMIX_NAMESPACE", "PMI_JOBID", "MASTER_PORT", "MASTER_ADDR")}
    env.update(kw)
    return env


def _publish(so, path, ident, env):
    code = ("import ctypes as C, sys; L = C.CDLL(sys.argv[1]); L.mk_idfile_publish.argtypes = [C.c_char_p, C.c_char_p]; "
            "sys.exit(L.mk_idfile_publish(sys.argv[2].encode(), bytes.fromhex(sys.argv[3])))")
    subprocess.check_call([sys.executable, "-c", code, so, path, ident.hex()], env=env)


def _wait(so, path, timeout, max_age, env):
    out = subprocess.run([sys.executable, "-c", WAITER, so, path, str(timeout), str(max_age)], env=env, capture_output=True, text=True, timeout=60)
    rc, hexid = out.stdout.split(" ")
    return int(rc), bytes.fromhex(hexid)


def test_reader_ignores_the_id_of_another_launch(tmp_path):
    L, so = _lib()
    path = str(tmp_path / "id")
    old, new = bytes([1]) * 256, bytes(range(256))
    _publish(so, path, old, _env(MK_LAUNCH_ID="job-41"))                    # leftover of a launch that died
    rc, _ = _wait(so, path, 0.4, 120, _env(MK_LAUNCH_ID="job-42"))
    assert rc == 2                                                          # seen, not taken
    # rank 0 of this launch publishes while the reader waits
    p = subprocess.Popen([sys.executable, "-c", WAITER, so, path, "20", "120"], env=_env(MK_LAUNCH_ID="job-42"), stdout=subprocess.PIPE, text=True)
    time.sleep(0.5)
    L.mk_idfile_clear(path.encode())
    _publish(so, path, new, _env(MK_LAUNCH_ID="job-42"))
    out, _ = p.communicate(timeout=60)
    rc, hexid = out.split(" ")
    assert int(rc) == 0 and bytes.fromhex(hexid) == new


def test_reader_without_a_launch_id_ignores_old_files(tmp_path):
    L, so = _lib()
    path = str(tmp_path / "id")
    _publish(so, path, bytes([7]) * 256, _env())
    past = time.time() - 3600
    os.utime(path, (past, past))                                            # an hour old: a leftover
    assert _wait(so, path, 0.4, 120, _env())[0] == 2
    _publish(so, path, bytes([9]) * 256, _env())                            # fresh file, same launch (no nonce either side)
    rc, ident = _wait(so, path, 5, 120, _env())
    assert rc == 0 and ident == bytes([9]) * 256
    # torchrun-style launch: MASTER_PORT is the nonce; a file without one is not this launch's
    assert _wait(so, path, 0.4, 120, _env(MASTER_ADDR="127.0.0.1", MASTER_PORT="29511"))[0] == 2


def test_missing_file_times_out(tmp_path):
    L, so = _lib()
    t0 = time.time()
    assert _wait(so, str(tmp_path / "nothing"), 0.3, 120, _env())[0] == 1
    assert time.time() - t0 < 30
