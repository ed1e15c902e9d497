"""Summarise rocprofv3 (rocpd sqlite) outputs of scripts/gpu_profile.sh into a text file for profiles/."""
import glob, os, sqlite3, sys

d = sys.argv[1]
out = []
db = glob.glob(os.path.join(d, "trace", "*.db"))
if db:
    c = sqlite3.connect(db[0])
    out.append("== rocprofv3 --kernel-trace --stats (per kernel: calls, avg us, min us, max us, total us)")
    q = ("select s.kernel_name, count(*), avg(d.end-d.start)/1e3, min(d.end-d.start)/1e3, max(d.end-d.start)/1e3, "
         "sum(d.end-d.start)/1e3, max(s.sgpr_count), max(s.arch_vgpr_count), max(d.group_segment_size), max(d.grid_size_x), max(d.workgroup_size_x) "
         "from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id=s.id group by s.kernel_name order by 6 desc")
    for r in c.execute(q):
        out.append("%-70s calls=%-4d avg=%12.1f min=%12.1f max=%12.1f total=%14.1f  sgpr=%s vgpr=%s lds=%s grid=%s wg=%s" % r)
    out.append("== the same, batch launches only (largest grid per kernel): calls, avg us")
    q2 = ("select s.kernel_name, d.grid_size_x, (d.end-d.start)/1e3 from rocpd_kernel_dispatch d "
          "join rocpd_info_kernel_symbol s on d.kernel_id=s.id")
    byk = {}
    for name, grid, us in c.execute(q2):
        byk.setdefault(name, []).append((grid, us))
    for name, l in sorted(byk.items()):
        if "tbc" in name:
            gmax = max(g for g, _ in l)
            big = [u for g, u in l if g == gmax]
            out.append("%-70s calls=%-4d avg=%12.1f" % (name, len(big), sum(big) / len(big)))
for p in sorted(glob.glob(os.path.join(d, "pmc*", "*.db"))):
    c = sqlite3.connect(p)
    out.append(f"== rocprofv3 --pmc ({os.path.basename(os.path.dirname(p))}): per kernel, counter per BATCH launch (mean over the largest-grid dispatches)")
    # the batch launches are the dispatches with the largest grid; bench.py's single-history time-to-verdict
    # checks launch the same kernels on one history and would drown the averages
    q = ("select s.kernel_name, i.name, d.id, sum(e.value), max(d.grid_size_x) from rocpd_pmc_event e "
         "join rocpd_info_pmc i on e.pmc_id=i.id join rocpd_kernel_dispatch d on e.event_id=d.event_id "
         "join rocpd_info_kernel_symbol s on d.kernel_id=s.id group by s.kernel_name, i.name, d.id order by 1,2,3")
    try:
        rows = {}
        for name, ctr, did, tot, grid in c.execute(q):
            rows.setdefault((name, ctr), []).append((grid, tot))
        for (name, ctr), l in sorted(rows.items()):
            if "tbc" not in name:
                continue
            gmax = max(g for g, _ in l)
            big = [v for g, v in l if g == gmax]
            out.append("%-60s %-24s batch_launches=%-3d grid=%-9d per_launch=%.6g" % (name[:60], ctr, len(big), gmax, sum(big) / len(big)))
    except Exception as e:
        out.append(f"   (query failed: {e})")
bl = [l for l in open(os.path.join(d, "trace.log")) if l.startswith('{"metric"')]
if bl:
    out.append("== bench.py line of the traced run")
    out.append(bl[-1].strip())
print("\n".join(out))
