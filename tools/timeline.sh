# Per-queue busy / idle accounting of the default (multi-stream, eager) bench step from a rocprofv3 kernel trace.
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/tl; rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -- python3 $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-kernel-timing > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections
rows = []
for f in glob.glob("/tmp/tl/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"] + " #g%sx%sx%s/s%s" % (int(r.get("Grid_Size_X",0))//max(int(r.get("Workgroup_Size_X",1)),1), r.get("Grid_Size_Y","?"), r.get("Grid_Size_Z","?"), r.get("Stream_Id", "?"))))
rows.sort()
import re
def short(n):
    m = re.search(r'(\w+_kernel|\w+Functor|copyBuffer|fillBuffer)', n)
    return m.group(1) if m else n[:30]
# step boundaries: adam_prep_kernel marks the start of an update
marks = [s for s, e, q, n in rows if "adam_prep" in n]
print("steps seen", len(marks))
a, b = marks[-3], marks[-2]
sel = [r for r in rows if a <= r[0] < b]
print("step length %.3f ms, kernels %d" % ((b - a) / 1e6, len(sel)))
byq = collections.defaultdict(list)
for r in sel: byq[r[2]].append(r)
for q, rs in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    busy = sum(e - s for s, e, _, _ in rs)
    span = rs[-1][1] - rs[0][0]
    gaps = [rs[i+1][0] - rs[i][1] for i in range(len(rs)-1)]
    small = [g for g in gaps if 0 <= g < 20000]
    print("queue %s: %d kernels, busy %.3f ms, span %.3f ms, gaps<20us: n=%d sum %.3f ms (median %.1f us), gaps>=20us sum %.3f ms" % (
        q, len(rs), busy/1e6, span/1e6, len(small), sum(small)/1e6, (sorted(small)[len(small)//2]/1e3 if small else 0), sum(g for g in gaps if g >= 20000)/1e6))
    kinds = collections.Counter()
    for s, e, _, n in rs: kinds[short(n)] += e - s
    print("   top:", [(k, round(v/1e6, 3)) for k, v in kinds.most_common(6)])
q1 = max(byq.items(), key=lambda kv: len(kv[1]))[1]
t0 = q1[0][0]
print("gaps >= 15 us on the main queue (at ms: gap us, before -> after):")
for i in range(len(q1) - 1):
    g = q1[i+1][0] - q1[i][1]
    if g >= 15000:
        others = [short(n) for s_, e_, q_, n in sel if q_ != q1[0][2] and s_ < q1[i+1][0] and e_ > q1[i][1]]
        print("  %.3f: %.1f  %s -> %s   | running elsewhere: %s" % ((q1[i][1] - t0) / 1e6, g / 1e3, short(q1[i][3]), short(q1[i+1][3]), others[:4]))
big = max(range(len(q1) - 1), key=lambda i: q1[i+1][0] - q1[i][1])
g0, g1 = q1[big][1], q1[big+1][0]
print("kernels around the largest main-queue gap (%.3f .. %.3f ms):" % ((g0 - t0) / 1e6, (g1 - t0) / 1e6))
for s_, e_, q_, n in sel:
    if e_ > g0 - 150000 and s_ < g1 + 400000:
        print("   q%s  %.3f .. %.3f  (%.1f us)  %s %s" % (q_, (s_ - t0) / 1e6, (e_ - t0) / 1e6, (e_ - s_) / 1e3, short(n), n[n.rfind("#"):]))
# union busy time over all queues
ev = sorted([(s, 1) for s, e, _, _ in sel] + [(e, -1) for s, e, _, _ in sel])
cur = 0; last = None; idle = 0; conc = collections.Counter()
for t, d in ev:
    if last is not None: conc[cur] += t - last
    cur += d; last = t
print("concurrency histogram (ms):", {k: round(v/1e6, 3) for k, v in sorted(conc.items())})
PY
