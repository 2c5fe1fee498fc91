"""The label-sparse decoders' fused cross-entropy alone at the step's shapes (fp16 logits [2432, 175104] for the entity head,
[2432, 29056] for the text head): us per launch and bytes per second, and the result against torch. Run under
STONK_HIP_LIB=ab_ref/libstonk_hip.so for another build of the library."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stonkgs_amd import _hip as hip  # noqa: E402

R = 2432
for ncols, npad in ((175094, 175104), (28996, 29056)):
    g = torch.Generator(device="cuda").manual_seed(1)
    logits = (torch.randn(R, npad, device="cuda", generator=g) * 3).to(torch.float16)
    tgt = torch.randint(0, ncols, (R,), device="cuda", generator=g, dtype=torch.int32)
    cnt = torch.tensor([R], dtype=torch.int32, device="cuda")
    loss = torch.zeros(1, device="cuda")
    d = torch.empty(R, npad, device="cuda", dtype=torch.bfloat16)
    err = torch.zeros(1, dtype=torch.int32, device="cuda")

    def run():
        hip.call("stonk_softmax_xent_f16_fwd_bwd", logits.data_ptr(), npad, ncols, npad, tgt.data_ptr(), cnt.data_ptr(),
                 loss.data_ptr(), d.data_ptr(), npad, 1.0, R, err.data_ptr(), hip.stream_ptr())

    run()
    torch.cuda.synchronize()
    x = logits[:, :ncols].float()
    ref_loss = torch.nn.functional.cross_entropy(x, tgt.long(), reduction="sum")
    p = torch.softmax(x[:64], -1)
    p[torch.arange(64), tgt[:64].long()] -= 1
    e_loss = abs(float(loss) - float(ref_loss)) / float(ref_loss)
    e_grad = float((d[:64, :ncols].float() - p / R).abs().max() * R)
    pad_zero = bool((d[:, ncols:] == 0).all())
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        run()
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 10 * 1e3
    print(f"{ncols} columns: {t:.0f} us, {R * npad * 4 / t / 1e6:.2f} TB/s of algorithmic bytes (fp16 in, bf16 out); loss rel err "
          f"{e_loss:.1e}, max |dlogit err| x rows {e_grad:.1e}, padding zero {pad_zero}, err flag {int(err)}", flush=True)
