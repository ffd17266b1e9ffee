"""Device time of one full selection (payload emission, no base, no metric passes) at Mistral dims."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import llama_cpp_amd  # noqa
from llama_cpp_amd import seeddelta, scan as SC
from llama_cpp_amd.dense import make_dense
seeddelta.init(0)
out = []
for kind, dims, Ks in (("ffn_gate", (4096, 14336), (128, 256, 384, 512, 640, 768)), ("ffn_down", (14336, 4096), (256, 512, 768, 1024, 1280, 1536))):
    W = SC.DenseWeight(make_dense(dims[0], dims[1], "q8_0", seed=1, structure=0.3))
    for base in (False, True):
        p = SC.ScanParams(scheme="coo", base=base, row_scale=base, eval_cols=0, eval_x=0)
        s = SC.ScanSession(1)
        r = s.tensor_kgrid(W, p, Ks, kind, 0, None)
        p.base_mode = 2 if base else 0                     # keep the fit: the selection alone is timed
        for rep in range(3):
            r = s.tensor_kgrid(W, p, Ks, kind, 0, None)
        out.append("%s%s %.3f ms" % (kind, " +base+rs" if base else "", sum(x["gpu_ms"] for x in r)))
    W.close()
lib = os.environ.get("SEEDDELTA_HIP_LIB")
print(lib.split("/")[-2] if lib else "default", " | ".join(out))
