import sys
sys.path.insert(0, ".")
import numpy as np
from claude_semantic_search_amd.flat_index import IndexFlatIP
from claude_semantic_search_amd import synth
cent = synth.rows(40, 768, 11)
cx = IndexFlatIP(768)
cx.add(np.repeat(cent, 6000, axis=0) + 0.01 * synth.rows(240_000, 768, 12), normalize=True)
cx.set_search_mode("coarse")
cq = cent[:24] + 0.005 * synth.rows(24, 768, 13)
res = []
for it in range(20):
    D, I = cx.search(cq, 10, normalize=True)
    res.append((D.copy(), I.copy(), cx.last_flagged(), cx.last_swept()))
cx.set_search_mode("exact_fp32")
De, Ie = cx.search(cq, 10, normalize=True)
for it, (D, I, f, s) in enumerate(res):
    print(it, "flagged", f, "swept", s, "max|D-D0|", float(np.abs(D - res[0][0]).max()), "ids differ from first", int((I != res[0][1]).sum()), "max|D-Dexact|", float(np.abs(D - De).max()), "ids differ from exact", int((I != Ie).sum()))
