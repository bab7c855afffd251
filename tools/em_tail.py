import numpy as np, sys
sys.path.insert(0, '.')
from rpvg_amd import engine as eng_mod, synth
from rpvg_amd.batch import make_params
b = synth.generate(**synth.FULL)
e = eng_mod.Engine(0)
model = sys.argv[1] if len(sys.argv) > 1 else "haplotype-transcripts"
est, _ = e.run(model, make_params(), e.prepare(b))
its = np.concatenate([np.asarray(x.em_iters) for x in est if len(x.em_iters)])
ncols = np.concatenate([[len(c) for c in x.em_cols] for x in est if len(x.em_iters)])
rows = np.diff(b.cluster_row_off.astype(np.int64))
rows_per = np.concatenate([[rows[k]] * len(x.em_iters) for k, x in enumerate(est) if len(x.em_iters)])
ents = np.diff(b.grp_idx_off[b.row_grp_off[b.cluster_row_off.astype(np.int64)].astype(np.int64)].astype(np.int64))
ents_per = np.concatenate([[ents[k]] * len(x.em_iters) for k, x in enumerate(est) if len(x.em_iters)])
print("problems", len(its), "total its", its.sum(), "max", its.max(), "p99", np.percentile(its, 99), "mean", its.mean())
order = np.argsort(-its)[:12]
for i in order:
    print(int(its[i]), int(ncols[i]), int(rows_per[i]), int(ents_per[i]))
big = np.argsort(-(its * rows_per))[:8]
print("by its*rows:")
for i in big:
    print(int(its[i]), int(ncols[i]), int(rows_per[i]))
