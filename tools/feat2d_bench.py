"""Times the stock-PyTorch 2-D feature extractor (fwd+bwd, both views) in NCHW vs channels_last."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stereo_toolbox_amd.models.GwcNet.gwcnet import feature_extraction
torch.backends.cudnn.benchmark = True
dev = torch.device("cuda:0")
for fmt in ("nchw", "nhwc"):
    m = feature_extraction(True, 12).to(dev).train()
    x = [torch.randn(1, 3, 576, 960, device=dev) for _ in range(2)]
    if fmt == "nhwc":
        m = m.to(memory_format=torch.channels_last)
        x = [t.contiguous(memory_format=torch.channels_last) for t in x]
    def step():
        outs = [m(t) for t in x]
        loss = sum(o["gwc_feature"].square().mean() + o["concat_feature"].square().mean() for o in outs)
        loss.backward()
    for _ in range(4): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print(fmt, "fwd+bwd both views: %.2f ms" % (dt * 1e3), flush=True)
    with torch.no_grad():
        m.eval()
        for _ in range(3): [m(t) for t in x]
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): [m(t) for t in x]
        torch.cuda.synchronize(); print(fmt, "eval fwd both views: %.2f ms" % ((time.perf_counter() - t0) / 10 * 1e3), flush=True)
