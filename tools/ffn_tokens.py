"""Gated FFN block (Mistral-7B layer, COO K=128 + base, SiLU) at several token counts: us per call of sdh_ffn_forward
on device buffers, 8 distinct layers rotating, timed with the library's span events on an explicit stream.
    python3 tools/ffn_tokens.py [T ...]        (SDH_FFN_KIND=block16 for block payloads, SDH_FFN_FLAGS=0x40 for the folded forms)"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B                                           # noqa: E402


def main():
    import torch
    from llama_cpp_amd import seeddelta
    from llama_cpp_amd.payload import config_payload
    seeddelta.init(0)
    dev = torch.device("cuda:0")
    Ts = [int(a) for a in sys.argv[1:]] or [1, 2, 3, 4, 8, 64, 512]
    kind = os.environ.get("SDH_FFN_KIND", "coo")
    flags = int(os.environ.get("SDH_FFN_FLAGS", "0"), 0)
    names = {"coo": ("cfg4_mistral_gate_coo_k128", "cfg4_mistral_down_coo_k128"),
             "block16": ("cfg4_mistral_gate_block16_k128", "cfg4_mistral_down_block16_k128")}[kind]
    layers = []
    for i in range(4):
        g = seeddelta.SeedDeltaWeight(config_payload(names[0], seed=100 + 3 * i))
        u = seeddelta.SeedDeltaWeight(config_payload(names[0], seed=101 + 3 * i))
        d = seeddelta.SeedDeltaWeight(config_payload(names[1], seed=102 + 3 * i))
        layers.append((g, u, d))
    st = torch.cuda.Stream(device=dev)
    out = {}
    for T in Ts:
        x = torch.randn((T, 4096), dtype=torch.float32, device=dev)
        y = torch.empty((T, 4096), dtype=torch.float32, device=dev)
        torch.cuda.synchronize()
        def fn():
            for g, u, d in layers:
                seeddelta.ffn_forward_device(g, u, d, seeddelta.ACT_SILU, x.data_ptr(), 4096, T, y.data_ptr(), 4096, flags, st.cuda_stream)
        reps = 40 if T <= 64 else 8
        ev_ms, host_ms = B.timed_on_stream(fn, reps, st, seeddelta, warm=3)
        out[T] = round(ev_ms / reps / len(layers) * 1e3, 2)
    print(json.dumps({"kind": kind, "flags": flags, "us_per_layer_by_tokens": out}))


if __name__ == "__main__":
    main()
