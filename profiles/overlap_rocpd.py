"""Concurrency of kernels from different queues / streams in a rocprofv3 rocpd database: wall span, summed kernel time, time with
>= 1 and >= 2 kernels in flight, and for one kernel name (default k_mbc_onchip) the share of its run time during which a kernel
of ANOTHER queue was running.  Usage: python profiles/overlap_rocpd.py DB [kernel_substring] [frac=0.5]"""
import sqlite3
import sys


def main(db, name="k_mbc_onchip", frac=0.5):
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(rocpd_kernel_dispatch)")]
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
    rows = c.execute(f"""select s.display_name, d.start, d.end, {qcol or 0} from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s
                         on d.kernel_id = s.id order by d.start""").fetchall()
    rows = rows[int(len(rows) * (1.0 - float(frac))):]
    ev = []
    for n, s, e, q in rows:
        ev.append((s, 1)); ev.append((e, -1))
    ev.sort()
    t1 = t2 = 0
    depth, last = 0, ev[0][0]
    for t, d in ev:
        if depth >= 1: t1 += t - last
        if depth >= 2: t2 += t - last
        depth += d; last = t
    span = rows[-1][2] - rows[0][1]
    busy = sum(e - s for _, s, e, _ in rows)
    queues = sorted({q for *_, q in rows})
    print(f"kernels {len(rows)} queues {queues} span_ms {span / 1e6:.2f} sum_kernel_ms {busy / 1e6:.2f} >=1_in_flight {t1 / span:.3f} >=2_in_flight {t2 / span:.3f}")
    tgt = [(s, e, q) for n, s, e, q in rows if name in n]
    oth = sorted((s, e, q) for n, s, e, q in rows)
    tot = ov = 0
    j0 = 0
    for s, e, q in tgt:
        tot += e - s
        cover = []
        for s2, e2, q2 in oth:
            if s2 >= e: break
            if e2 <= s or q2 == q: continue
            cover.append((max(s, s2), min(e, e2)))
        cover.sort(); cur = s
        for a, b in cover:
            if b > cur:
                ov += b - max(a, cur); cur = b
    if tgt:
        print(f"{name}: {len(tgt)} launches, {tot / 1e6:.2f} ms, share with a kernel of another queue in flight {ov / max(tot, 1):.3f}")


if __name__ == "__main__":
    main(*sys.argv[1:4])
