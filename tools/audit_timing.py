"""Device time of a run in audit mode (tp_set_audit: the audit row's tag lists kept by the wave stages, or by the thread-per-file
kernel for the files they decline, and emitted as text) beside the same run without it, and the host time of the C calls
behind a whole audit pass: tp_run_lines (copies included), tp_format_audit_rows, tp_format_review_rows - timed at the C ABI,
without the Python mirror's per-file slicing."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagecaptioner_amd import Config, TagPipe  # noqa: E402
from imagecaptioner_amd._abi import tp_result  # noqa: E402
from imagecaptioner_amd.corpus import make_corpus  # noqa: E402

name, n = sys.argv[1], int(sys.argv[2])
c = make_corpus(name, n, seed=1)
p = TagPipe(Config(**c.config_kwargs))
L, h = p._L, p._h
arena = np.ascontiguousarray(c.arena)
offsets = np.ascontiguousarray(c.offsets, dtype=np.uint64)


def run(audit):
    res = tp_result()
    L.tp_set_audit(h, int(audit))
    t0 = time.perf_counter()
    rc = L.tp_run_lines(h, arena.ctypes.data, offsets.ctypes.data, n, C.byref(res))
    dt = time.perf_counter() - t0
    L.tp_set_audit(h, 0)
    assert rc == 0, L.tp_last_error()
    return res, dt


for _ in range(2):
    res, plain_wall = run(False)
    plain = p.last_timing()
    L.tp_release(h, C.byref(res))
names = [b"%08d.png" % i for i in range(n)]
noffs = np.zeros(n + 1, dtype=np.uint64)
noffs[1:] = np.cumsum([len(b) for b in names])
narena = np.frombuffer(b"".join(names) + b"\0", dtype=np.uint8).copy()
for _ in range(2):
    res, run_wall = run(True)
    tm = p.last_timing()
    roffs = np.zeros(n + 1, dtype=np.uint64)
    need = C.c_uint64()
    t0 = time.perf_counter()
    L.tp_format_audit_rows(h, C.byref(res), narena.ctypes.data, noffs.ctypes.data, b"top_tags.txt", b"sgrui", None, 0, roffs.ctypes.data, C.byref(need))
    buf = np.empty(max(need.value, 1), dtype=np.uint8)
    rc = L.tp_format_audit_rows(h, C.byref(res), narena.ctypes.data, noffs.ctypes.data, b"top_tags.txt", b"sgrui", buf.ctypes.data, need.value, roffs.ctypes.data, C.byref(need))
    fmt_wall = time.perf_counter() - t0
    assert rc == 0
    t0 = time.perf_counter()
    need2 = C.c_uint64()
    L.tp_format_review_rows(h, C.byref(res), 0, narena.ctypes.data, noffs.ctypes.data, narena.ctypes.data, noffs.ctypes.data, b"primary=0.35", b"top_tags.txt", None, 0,
                            roffs.ctypes.data, C.byref(need2))
    rev_wall = time.perf_counter() - t0
    L.tp_release(h, C.byref(res))
rows = need.value
print("%s %d files: plain run device %.2f ms (%.1f M files/s), tp_run_lines %.3f s; audit run device %.2f ms (lean front %.2f, wave stages in all %.2f, general kernel %.2f ms on %d "
      "files, emit %.2f ms, other incl. audit text %.2f ms) = %.2f M files/s; audit.csv %.1f MB (%.0f B/row); at the C ABI: tp_run_lines in audit mode %.3f s + "
      "tp_format_audit_rows (sizes, then text) %.3f s = %.3f s for the audit pass; tp_format_review_rows (sizes only) %.3f s" % (
          name, n, plain["total_ms"], n / plain["total_ms"] / 1e3, plain_wall, tm["total_ms"], tm["lean_front_ms"], tm["select_ms"], tm["general_ms"], tm["general_files"],
          tm["emit_ms"], tm["other_ms"], n / tm["total_ms"] / 1e3, rows / 1e6, rows / n, run_wall, fmt_wall, run_wall + fmt_wall, rev_wall))
