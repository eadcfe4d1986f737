"""Summarise a rocprofv3 results .db: kernels grouped by (name, grid, lds) in order of first appearance.
usage: kstats.py results.db [name-filter]"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
rows = db.execute("select name, grid_x, grid_y, grid_z, workgroup_x, lds_size, start, end from kernels order by start").fetchall()
groups, order = {}, []
for name, gx, gy, gz, wx, lds, s, e in rows:
    short = re.sub(r"\(anonymous namespace\)::", "", name)
    short = re.sub(r"\(.*", "", short).replace("void ", "")
    key = (short, gx // max(wx, 1), gy, gz, lds)
    if key not in groups:
        groups[key] = []
        order.append(key)
    groups[key].append(e - s)
for key in order:
    if flt and flt not in key[0]:
        continue
    d = sorted(groups[key])
    med = d[len(d) // 2]
    print(f"{key[0][:58]:58s} grid {key[1]:5d}x{key[2]:3d}x{key[3]:4d} lds {key[4]:6d}  n={len(d):3d}  med {med/1e3:8.1f} us  min {d[0]/1e3:8.1f}")
