"""Where inside a step a wave of fused_dna_eval_kernel waits.

Runs one evaluator launch of a bench workload on an ABLATION library built with
-DRDAMD_ABL_STAMPS=<mask> (csrc/kernels_fused.hip: s_memtime readings at the phase boundaries of
RDAMD_STEP; profiles/step_timeline.sh builds the libraries and puts each in the product library's
place in turn) and turns the readings of the stamped waves into cycles per phase and step kind.

  python profiles/step_timeline.py --config c2 [--job 5 --stride 12] [--json out.json]

Phases (readings 0..4, see the kernel): 0->1 wait for the operand tables (LDS-DMA, vmcnt(0)),
1->2 table rows back from LDS + the next step's requests issued, 2->3 the matrix-vector
product(s), 3->4 the product with the sibling / rescale test (+ park), 4->0' loop edge.
A reading that was not taken (mask) is 0: its phase merges into the next one that was.
"""
import argparse
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import root_digger_amd as rd
from root_digger_amd import synth

SHAPES = {   # the bench's shapes (bench.py CONFIGS and the shard shapes of profiles/r5_shard_summary.txt)
    "c2": dict(n=100, S=50_000, seed_index=1, batch=197),
    "c4s": dict(n=500, S=62_500, seed_index=3, batch=13),      # c4 / 8: one GPU's site shard
    "c5s": dict(n=1000, S=50_000, seed_index=4, batch=13),     # c5 4 x 2: one GPU's shard
    "c2s": dict(n=100, S=6_250, seed_index=1, batch=197),      # c2 / 8
}
KINDS = {0: "TT", 1: "RT", 2: "RP"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c2", choices=sorted(SHAPES))
    ap.add_argument("--job", type=int, default=5)
    ap.add_argument("--stride", type=int, default=None, help="every stride-th workgroup of the job stamps (64 at most)")
    ap.add_argument("--json", default=None)
    ap.add_argument("--label", default="")
    ap.add_argument("--repeat-classes", type=int, default=None, help="class limit of the pseudo-tips (16: the 16-row kernel)")
    ap.add_argument("--rescale-speculation", type=int, default=-1)
    args = ap.parse_args()
    if not hasattr(rd.lib, "rdamd_abl_stamps_config"):
        raise SystemExit("the library in place was not built with -DRDAMD_ABLATION -DRDAMD_ABL_STAMPS=<mask>")
    rd.lib.rdamd_abl_stamps_config.argtypes = [C.c_uint, C.c_uint]
    rd.lib.rdamd_abl_stamps_read.argtypes = [C.c_void_p]
    sh = SHAPES[args.config]
    n, S, R, K = sh["n"], sh["S"], 4, 4
    w = synth.workload(n, S, K, R, 0xD166E5 + sh["seed_index"])
    tree = rd.Tree.from_newick(w["newick"])
    part = rd.Partition.for_tree(tree, K, S, R, attributes=rd.ATTRIB_SITE_REPEATS)
    if args.repeat_classes is not None:
        part.set_site_repeats(args.repeat_classes)
    part.set_rescale_speculation(args.rescale_speculation)
    for label, seq in w["seqs"].items():
        part.set_tip_states(tree.tip_index(label), rd.MAP_NT, seq)
    freqs = part.empirical_frequencies()
    part.set_frequencies(0, freqs)
    part.set_category_rates(w["rates"])
    rng = np.random.default_rng(7)
    nb = sh["batch"]
    roots = [tree.root_location(int(i) % tree.root_count()) for i in range(nb)]
    scheds = [part.schedule(*tree.generate_operations(rl)) for rl in roots]
    subst = np.array([synth.random_params(12, rng) for _ in range(nb)])
    fr = np.tile(np.asarray(freqs), (nb, 1))
    blocks = (S + 127) // 128                      # workgroups per job at two sites per lane
    stride = args.stride or max(1, blocks // 48)
    job = min(args.job, nb - 1)
    for rep in range(3):                           # warm-up, then the measured launch (the buffer is zeroed by _config)
        part.profile_enable(rep == 2)
        assert rd.lib.rdamd_abl_stamps_config(job if rep == 2 else 0xFFFFFFFF, stride) == 1
        part.evaluate_batch(scheds, subst, fr)
        part.sync()
    prof = part.profile_read()
    kernel_ms = prof["fused"][0] / max(1, prof["fused"][1])
    buf = np.zeros((64, 4096, 8), dtype=np.uint32)
    assert rd.lib.rdamd_abl_stamps_read(buf.ctypes.data_as(C.c_void_p)) == 1
    steps_per_eval = scheds[job].stats()["steps"]
    t = buf.astype(np.int64)
    waves = [wv for wv in range(64) if t[wv, 0, 0] or t[wv, 0, 1]]
    out = {"config": args.config, "label": args.label, "kernel_ms": kernel_ms, "stamped_waves": len(waves),
           "steps_per_rate_pass": int(steps_per_eval), "job": job, "stride": stride}
    if not waves:
        raise SystemExit("no stamped wave: " + json.dumps(out))
    per_kind = {k: [] for k in KINDS}
    wave_span, step_total = [], []
    for wv in waves:
        rec = t[wv]
        nsteps = int(np.count_nonzero(rec[:, 0]))
        rec = rec[:nsteps]
        t0 = rec[:, 0] | (rec[:, 1] << 32)
        lo0 = rec[:, 0]
        # the later readings are low words: differences modulo 2^32 against reading 0
        def since0(col):
            d = (rec[:, col] - lo0) & 0xFFFFFFFF
            return np.where(rec[:, col] == 0, -1, d)
        r1, r2, r3, r4 = since0(2), since0(3), since0(4), since0(5)
        kind = rec[:, 6] & 3
        nxt = np.append(t0[1:] - t0[:-1], -1)
        for k in KINDS:
            sel = (kind == k) & (nxt >= 0) & (nxt < 1 << 22)     # (not the step a rate pass ends with)
            per_kind[k].append(np.stack([r1[sel], r2[sel], r3[sel], r4[sel], nxt[sel]], axis=1))
        wave_span.append(int(t0[-1] - t0[0]))
        step_total.append(nsteps)
    out["steps_recorded_per_wave"] = int(np.median(step_total))
    out["wave_span_ticks_median"] = int(np.median(wave_span))
    table = {}
    for k, name in KINDS.items():
        a = np.concatenate(per_kind[k]) if per_kind[k] else np.zeros((0, 5))
        if not len(a):
            continue
        prev = np.zeros(len(a))
        row = {"steps": int(len(a))}
        names = ["tables landed", "rows back, next requested", "matvec issued", "step done", "next step's top"]
        for c, nm in enumerate(names):
            taken = a[:, c] >= 0
            if not taken.any():
                continue
            d = a[:, c] - prev
            row[nm] = {"mean": float(d[taken].mean()), "p50": float(np.median(d[taken])), "p90": float(np.percentile(d[taken], 90))}
            prev = np.where(taken, a[:, c], prev)
        row["whole step"] = {"mean": float(a[:, 4].mean()), "p50": float(np.median(a[:, 4])), "p90": float(np.percentile(a[:, 4], 90))}
        table[name] = row
    out["ticks"] = table
    # ticks -> time: the stamped waves' lifetime against the launch
    all_steps = sum(table[k]["steps"] * table[k]["whole step"]["mean"] for k in table)
    out["mean_ticks_per_step"] = all_steps / max(1, sum(table[k]["steps"] for k in table))
    print("%s %s: kernel %.4f ms, %d stamped waves, %d steps each, %.0f ticks per step on average" %
          (args.config, args.label, kernel_ms, len(waves), out["steps_recorded_per_wave"], out["mean_ticks_per_step"]))
    for k in table:
        row = table[k]
        print("  %s (%d steps): " % (k, row["steps"]) +
              "  ".join("%s %.0f (p50 %.0f, p90 %.0f)" % (nm, v["mean"], v["p50"], v["p90"])
                        for nm, v in row.items() if nm != "steps"))
    if args.json:
        with open(args.json, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
