import sys, numpy as np, time
import os; sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import run_l2sim as R
kernels = R.pkg("kernels")
z = np.load(sys.argv[1] if len(sys.argv) > 1 else '/tmp/gather_part.npz')
rowptr, col = z["rowptr"], np.ascontiguousarray(z["col"], dtype=np.int32)
tasks, fix, nslots, seg = kernels.build_plan(rowptr, 1024, z["slice_cnt"], 96, row_flags=z["row_flags"])
seg = [int(seg[i]) for i in range(9)]
kbeg = (tasks[:, 0].astype(np.int64) & 0xffffffff) | (tasks[:, 1].astype(np.int64) << 32)
t2 = np.stack([kbeg, tasks[:, 2].astype(np.int64)], 1)
by_slice = [t2[seg[s]:seg[s + 1]] for s in range(8)]
L = R.load_sim()
def rep(name, tb, c, window, passes, lpp, **kw):
    h, m = R.simulate(L, tb, c, window, passes, lpp, **kw)
    print("%-64s hit %5.1f %%  %.2f GB  tasks %d" % (name, 100.0*h/(h+m), m*128/1e9, sum(t.shape[0] for t in tb)), flush=True)
# 16 time slices: split every task by (col // 8) % 2, halves run one after the other on the XCD
def split_tasks(tb, col, parts):
    newcol = col.copy()
    out = [[[] for _ in range(parts)] for _ in range(8)]
    for s, t in enumerate(tb):
        for kb, ln in t:
            seg_c = col[kb:kb+ln]
            key = (seg_c // 8) % parts
            order = np.argsort(key, kind="stable")
            newcol[kb:kb+ln] = seg_c[order]
            cnt = np.bincount(key, minlength=parts)
            off = kb
            for p in range(parts):
                if cnt[p]:
                    out[s][p].append((off, cnt[p]))
                off += cnt[p]
    return [np.concatenate([np.array(x, dtype=np.int64).reshape(-1, 2) for x in out[s]]) for s in range(8)], newcol
t0 = time.time()
for parts in (2, 4):
    tb2, col2 = split_tasks(by_slice, col, parts)
    print("split into %d time slices per XCD: %.0f s" % (parts, time.time() - t0))
    rep("whole rows, %d time slices per XCD" % parts, tb2, col2, 1536, 1, 4)
    rep("64-feature passes one after the other, %d time slices" % parts, tb2, col2, 3072, 2, 2)
# unsliced tasks kept in (slice, length) order but window variants / chunk
for chunk in (256, 512, 2048):
    tk, fx, ns, sg = kernels.build_plan(rowptr, chunk, z["slice_cnt"], 96, row_flags=z["row_flags"])
    sg = [int(sg[i]) for i in range(9)]
    kb = (tk[:, 0].astype(np.int64) & 0xffffffff) | (tk[:, 1].astype(np.int64) << 32)
    tt = np.stack([kb, tk[:, 2].astype(np.int64)], 1)
    rep("whole rows, chunk %d" % chunk, [tt[sg[s]:sg[s+1]] for s in range(8)], col, 1536, 1, 4)
