"""A soak of the drop-in path: two threads, a model each (d_model 512 and 1024), greedy and beam resident passes over recycled
device buffers for a few minutes, every result checked against an eager pass computed up front; counts what ran, what did not
run as a resident launch (a batch beyond its rows, or a launch that timed out at a hand-off and was decoded again) and what came
back wrong.   python tools/soak.py [seconds] [--big] [--filler]"""
import os
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from care_amd import get_framework, get_translator
from care_amd.configs import feat_shapes, make_opt
from care_amd.synth import synth_state_dict

SECONDS = float(next((a for a in sys.argv[1:] if not a.startswith("-")), "120"))
stats = {}


def worker(name, config, B, seed):
    opt1, opt5 = make_opt(config, beam_size=1), make_opt(config, beam_size=5)
    model = get_framework(opt1).eval()
    model.load_state_dict(synth_state_dict(seed, [(k, tuple(v.shape)) for k, v in model.state_dict().items()],
                                           row_scale={"cls_head.tgt_word_prj.weight": {3: 6.0}}), strict=True)
    model.set_compute_dtype("fp16")
    model.to("cuda:0")
    tr = {1: get_translator(opt1), 5: get_translator(opt5)}
    gen = torch.Generator().manual_seed(seed)
    host = [torch.randn(s, generator=gen) for s in feat_shapes(opt1, 8 * B)]
    want = {}
    for k in range(8):
        dev = {"feats": [f[k * B: (k + 1) * B].to("cuda:0") for f in host]}
        for bm in (1, 5):
            want[(k, bm)] = tr[bm].translate_batch([model], dev, use_graph=False)
    st = stats[name] = dict(passes=0, wrong=0, errors=0, not_resident=0)
    t_end = time.time() + SECONDS
    i = 0
    while time.time() < t_end:
        k, bm = i % 8, (1, 5)[(i // 8) % 2]
        dev = {"feats": [f[k * B: (k + 1) * B].to("cuda:0") for f in host]}
        try:
            got = tr[bm].translate_batch([model], dev)
            st["wrong"] += int(got != want[(k, bm)])
            st["not_resident"] += int(not model.engine().last_decode.get("resident"))
        except Exception as exc:   # noqa: BLE001
            st["errors"] += 1
            st["last_error"] = repr(exc)[:200]
        st["passes"] += 1
        i += 1
        del dev


threads = [threading.Thread(target=worker, args=("d512", "msrvtt_care", 128, 3)),
           threading.Thread(target=worker, args=("d1024", "vatex_care_large", 32, 4))]
if "--big" in sys.argv:   # a third thread with batches of the segmented large-batch forms (early exit: host waits between segments)
    threads.append(threading.Thread(target=worker, args=("big", "msrvtt_base_ami", 1536, 5)))
stop = threading.Event()


def filler():   # keeps the CUs busy on a stream of its own: resident launches must share the chip (tests/test_gpu_resident.py)
    with torch.cuda.stream(torch.cuda.Stream()):
        a = torch.randn(4096, 4096, device="cuda:0")
        while not stop.is_set():
            for _ in range(20):
                a = (a @ a).clamp_(-1.0, 1.0)
            torch.cuda.current_stream().synchronize()


fill = threading.Thread(target=filler) if "--filler" in sys.argv else None
t0 = time.time()
if fill is not None:
    fill.start()
[t.start() for t in threads]
[t.join() for t in threads]
stop.set()
if fill is not None:
    fill.join()
print("soak %.0f s:" % (time.time() - t0), stats)
