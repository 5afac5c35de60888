"""Per-kernel mean of PMC counters from a rocprofv3 rocpd database (--pmc run)."""
import sqlite3
import sys


def main(db):
    c = sqlite3.connect(db)
    q = """select s.display_name, p.name, count(*), avg(e.value)
           from rocpd_pmc_event e join rocpd_info_pmc p on e.pmc_id = p.id
           join rocpd_kernel_dispatch d on e.event_id = d.event_id
           join rocpd_info_kernel_symbol s on d.kernel_id = s.id
           group by s.display_name, p.name order by 1, 2"""
    for name, ctr, n, avg in c.execute(q):
        short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:70]
        print(f"{short},{ctr},{n},{avg:.1f}")


if __name__ == "__main__":
    main(sys.argv[1])
