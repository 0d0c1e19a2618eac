#!/usr/bin/env python3
"""Per-kernel totals from a rocprofv3 --kernel-trace sqlite result (rocpd schema) -> text table (like --stats).
usage: python tools/trace_report.py gpurun_out/prof/bench_results.db [--csv out.csv]"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    con = sqlite3.connect(db)
    cur = con.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    rows = cur.execute(f"""select s.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start),
                           max(d.end-d.start) from '{disp}' d join '{sym}' s on s.id = d.kernel_id
                           group by s.kernel_name order by 3 desc""").fetchall()
    total = sum(r[2] for r in rows)
    lines = ["%-90s %8s %12s %10s %10s %10s %6s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct")]
    for kn, n, tot, avg, mn, mx in rows:
        lines.append("%-90s %8d %12.1f %10.2f %10.2f %10.2f %6.2f" % (kn[:90], n, tot / 1e3, avg / 1e3, mn / 1e3, mx / 1e3,
                                                                  100.0 * tot / total))
    lines.append("TOTAL kernel time %.3f ms over %d dispatches" % (total / 1e6, sum(r[1] for r in rows)))
    if "--sequence" in sys.argv:
        # the last N dispatches in launch order (N = the argument): one steady-state replay, kernel by kernel
        n_last = int(sys.argv[sys.argv.index("--sequence") + 1])
        seq = cur.execute(f"""select s.kernel_name, d.start, d.end from '{disp}' d join '{sym}' s on s.id = d.kernel_id
                               order by d.start desc limit {n_last}""").fetchall()[::-1]
        lines.append("last %d dispatches in order (us; gap = idle time since the previous kernel ended):" % len(seq))
        prev = None
        for kn, st, en in seq:
            short = kn.split("(")[0][-70:]
            lines.append("  %-70s %10.2f  gap %8.2f" % (short, (en - st) / 1e3, 0.0 if prev is None else (st - prev) / 1e3))
            prev = en
        lines.append("  span %.2f us, busy %.2f us" % ((seq[-1][2] - seq[0][1]) / 1e3, sum(e - s_ for _, s_, e in seq) / 1e3))
    txt = "\n".join(lines)
    print(txt)
    if "--out" in sys.argv:
        open(sys.argv[sys.argv.index("--out") + 1], "w").write(txt + "\n")


if __name__ == "__main__":
    main()
