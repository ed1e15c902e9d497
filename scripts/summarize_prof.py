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
for p in sorted(glob.glob(os.path.join(d, "pmc*", "*.db"))):
    c = sqlite3.connect(p)
    out.append(f"== rocprofv3 --pmc ({os.path.basename(os.path.dirname(p))}): per kernel, counter summed over dispatches / per dispatch average")
    q = ("select s.kernel_name, i.name, count(distinct d.id), sum(e.value) from rocpd_pmc_event e "
         "join rocpd_info_pmc i on e.pmc_id=i.id join rocpd_kernel_dispatch d on e.event_id=d.event_id "
         "join rocpd_info_kernel_symbol s on d.kernel_id=s.id group by s.kernel_name, i.name order by 1,2")
    try:
        for name, ctr, nd, tot in c.execute(q):
            if "tbc" in name:
                out.append("%-60s %-24s dispatches=%-3d sum=%.6g per_dispatch=%.6g" % (name[:60], ctr, nd, tot, tot / max(nd, 1)))
    except Exception as e:
        out.append(f"   (query failed: {e})")
bl = [l for l in open(os.path.join(d, "trace.log")) if l.startswith('{"metric"')]
if bl:
    out.append("== bench.py line of the traced run")
    out.append(bl[-1].strip())
print("\n".join(out))
