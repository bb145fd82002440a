"""A/B of the two focus pipelines on fuzzed BAMs (GPU box): slot pipeline (default) vs fused off
(MKP_FUSED=0).  Prints the first differing rows of every case so one gpurun call says where a divergence starts."""
import os, sys, tempfile
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import modkit_amd
from bamfuzz import Fuzz

CASES = [("m", 400, 3000), ("hm_comb", 400, 3000), ("hm_split", 400, 3000), ("implicit", 300, 3000), ("duplex", 300, 3000), ("hma", 300, 3000), ("hm_split", 300, 30000)]
FLAGS = [["--cpg", "--ref", "{fa}", "--filter-threshold", "0.7", "--force-allow-implicit"], ["--preset", "traditional", "--ref", "{fa}", "--filter-threshold", "0.66", "--force-allow-implicit"],
         ["--include-bed", "{bed}", "--filter-threshold", "0.7", "--force-allow-implicit"]]
bad = 0
with tempfile.TemporaryDirectory() as td:
    for ci, (prof, n, ml) in enumerate(CASES):
        contigs = (("ctgL", 90000),) if ml > 10000 else (("ctgA", 12000), ("ctgB", 3000))
        bam, fa, bed = Fuzz(300 + ci, contigs=contigs, profile=prof, n_reads=n, mean_len=ml).write(os.path.join(td, "c%d" % ci), bed=True)
        for fi, fl in enumerate(FLAGS):
            flags = [f.format(fa=fa, bed=bed) for f in fl]
            outs = {}
            for name, env in (("slots", {}), ("cover", {"MKP_FUSED": "0"})):
                for k in ("MKP_FUSED",): os.environ.pop(k, None)
                os.environ.update(env)
                o = os.path.join(td, "o_%s.bed" % name)
                try:
                    modkit_amd.pileup([bam, o] + flags); outs[name] = open(o).read().splitlines()
                except Exception as e:  # noqa
                    outs[name] = ["ERROR %s" % e]
            ref = outs["tiles"]
            for name in ("slots", "cover"):
                a = outs[name]
                if a == ref: continue
                bad += 1
                print("DIFF case %d (%s) flags %d pipeline %s: %d vs %d rows" % (ci, prof, fi, name, len(a), len(ref)))
                shown = 0
                for i in range(max(len(a), len(ref))):
                    x = a[i] if i < len(a) else "<none>"; y = ref[i] if i < len(ref) else "<none>"
                    if x != y:
                        print("  row %d\n   got  %s\n   want %s" % (i, x, y)); shown += 1
                        if shown >= 4: break
            print("case %d (%s) flags %d: rows %d %s" % (ci, prof, fi, len(ref), "ok" if outs["slots"] == ref and outs["cover"] == ref else "MISMATCH"))
print("slotdiff: %d mismatching runs" % bad)
