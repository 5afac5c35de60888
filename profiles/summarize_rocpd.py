"""Per-kernel statistics from a rocprofv3 rocpd SQLite database (`rocprofv3 --kernel-trace --stats`
writes `<name>_results.db` on this image).  Usage: python profiles/summarize_rocpd.py DB [OUT.csv]"""
import sqlite3
import sys


def main(db, out=None):
    c = sqlite3.connect(db)
    rows = c.execute(
        """select s.display_name, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start),
                  max(s.arch_vgpr_count), max(s.sgpr_count), max(d.grid_size_x*d.grid_size_y*d.grid_size_z/(d.workgroup_size_x*d.workgroup_size_y*d.workgroup_size_z))
           from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id
           group by s.display_name order by 3 desc"""
    ).fetchall()
    total = sum(r[2] for r in rows) or 1
    lines = ["kernel,calls,total_ms,avg_us,min_us,max_us,pct,vgpr,sgpr,max_workgroups"]
    for name, n, tot, avg, mn, mx, vg, sg, wgs in rows:
        short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:90].replace(",", ";")
        lines.append(f"{short},{n},{tot/1e6:.3f},{avg/1e3:.2f},{mn/1e3:.2f},{mx/1e3:.2f},{100*tot/total:.1f},{vg},{sg},{wgs}")
    text = "\n".join(lines)
    if out:
        open(out, "w").write(text + "\n")
    print(text)


if __name__ == "__main__":
    main(*sys.argv[1:3])
