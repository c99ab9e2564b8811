"""Prints the kernel timeline (start, duration, queue, name) of the last train step found in a
rocprofv3 --kernel-trace database. Usage: python benchmarks/timeline.py results.db [ms_window]"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
win = float(sys.argv[2]) if len(sys.argv) > 2 else 45.0
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
sym = [t for t in tabs if "kernel_symbol" in t][0]
dis = [t for t in tabs if "kernel_dispatch" in t][0]
cols = [r[1] for r in c.execute(f"pragma table_info({dis})")]
qcol = "queue_id" if "queue_id" in cols else "stream_id"
rows = c.execute(f"select d.start, d.end, d.{qcol}, k.kernel_name from {dis} d join {sym} k on d.kernel_id = k.id "
                 "order by d.start").fetchall()
t_end = rows[-1][1]
rows = [r for r in rows if r[0] >= t_end - win * 1e6]
t0 = rows[0][0]
for st, en, q, name in rows:
    short = name.split("(")[0].split("<")[0][-60:]
    print(f"{(st - t0) / 1e6:8.3f} ms  +{(en - st) / 1e6:7.3f}  q{q}  {short}")
