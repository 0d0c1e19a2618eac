#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc sqlite outputs (rocpd schema): per kernel, mean of every counter over dispatches.
usage: python tools/pmc_report.py gpurun_out/pmc*/pmc_results.db [--kernel contract]"""
import glob
import sqlite3
import sys


def report(db, kfilter):
    con = sqlite3.connect(db)
    cur = con.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    def tab(prefix):
        return [t for t in tabs if t.startswith(prefix)][0]
    ev, info, disp, sym = tab("rocpd_pmc_event"), tab("rocpd_info_pmc"), tab("rocpd_kernel_dispatch"), tab("rocpd_info_kernel_symbol")
    cols = [c[1] for c in cur.execute(f"pragma table_info('{disp}')")]
    rows = cur.execute(f"""
        select s.kernel_name, i.name, avg(v), count(*), avg(dur) from (
          select d.kernel_id as kid, e.pmc_id as pid, sum(e.value) as v, (d.end - d.start) as dur
          from '{ev}' e join '{disp}' d on e.event_id = d.{'event_id' if 'event_id' in cols else 'id'}
          group by d.id, e.pmc_id) q
        join '{sym}' s on s.id = q.kid join '{info}' i on i.id = q.pid
        group by s.kernel_name, i.name""").fetchall()
    out = {}
    for kn, cn, v, n, dur in rows:
        if kfilter in kn:
            out.setdefault(kn, {"_n": n, "_dur_us": dur / 1e3})[cn] = v
    return out


if __name__ == "__main__":
    kf = "contract"
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    if "--kernel" in sys.argv:
        kf = sys.argv[sys.argv.index("--kernel") + 1]
        args = [a for a in args if a != kf]
    for pat in args:
        for db in sorted(glob.glob(pat)):
            try:
                r = report(db, kf)
            except Exception as e:  # noqa
                print(db, "ERR", e)
                continue
            for kn, d in r.items():
                print("%s  [%s] dispatches=%d avg %.1f us" % (db, kn[:70], d.pop("_n"), d.pop("_dur_us")))
                for c, v in sorted(d.items()):
                    print("    %-28s %.4g" % (c, v))
