#!/usr/bin/env python3
"""Turn a rocprofv3 (ROCm 7.2) results .db into the small text summaries committed under profiles/.
usage: rocprof_summary.py <results.db> <out.md> [title]"""
import sqlite3
import sys


def main():
    db, out = sys.argv[1], sys.argv[2]
    title = sys.argv[3] if len(sys.argv) > 3 else db
    c = sqlite3.connect(db)
    rows = c.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    with open(out, "w") as f:
        f.write("# %s\n\nrocprofv3 --kernel-trace --stats (durations in microseconds)\n\n" % title)
        f.write("| kernel | calls | total us | avg us | % |\n|---|---|---|---|---|\n")
        for name, calls, total, avg, pct in rows:
            if pct < 0.05:
                continue
            f.write("| `%s` | %d | %.1f | %.1f | %.2f |\n" % (name[:110], calls, total, avg, pct))
        try:
            cols = [d[1] for d in c.execute("pragma table_info(kernels)")]
            want = [x for x in ("name", "vgpr_count", "accum_vgpr_count", "sgpr_count", "lds_size", "scratch_size",
                                "workgroup_size", "grid_size") if x in cols]
            seen = set()
            f.write("\n| kernel | " + " | ".join(want[1:]) + " |\n|" + "---|" * len(want) + "\n")
            for r in c.execute("select %s from kernels" % ",".join(want)):
                if r[0] in seen or not r[0].startswith(("void kws", "kws")):
                    continue
                seen.add(r[0])
                f.write("| `%s` | " % r[0][:60] + " | ".join(str(x) for x in r[1:]) + " |\n")
        except Exception as e:  # noqa
            f.write("\n(no per-dispatch resource table: %s)\n" % e)


if __name__ == "__main__":
    main()
