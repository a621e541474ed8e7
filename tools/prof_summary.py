"""Summarise a rocprofv3 --kernel-trace output (rocpd sqlite .db) into a per-kernel table (like --stats)."""
import glob, re, sqlite3, sys
path = sys.argv[1]
db = path if path.endswith(".db") else sorted(glob.glob(path + "/**/*.db", recursive=True))[0]
c = sqlite3.connect(db)
rows = c.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start), max(vgpr_count), max(lds_size), max(scratch_size) "
                 "from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
print(f"# source: {db}\n# total kernel time {tot/1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches")
print(f"{'total_ms':>10} {'pct':>6} {'calls':>6} {'avg_us':>10} {'min_us':>9} {'max_us':>9} {'vgpr':>5} {'lds':>7} {'scr':>5}  name")
for n, cnt, s, avg, mn, mx, vg, lds, scr in rows:
    n = re.sub(r"eagle::", "", n).replace("(ConvArgs)", "")
    print(f"{s/1e6:10.3f} {100*s/tot:6.2f} {cnt:6d} {avg/1e3:10.2f} {mn/1e3:9.2f} {mx/1e3:9.2f} {vg:5d} {lds:7d} {scr:5d}  {n[:110]}")
