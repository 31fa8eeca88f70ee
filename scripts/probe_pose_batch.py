import os, sys, time, torch
sys.path.insert(0, "/root/repo/dynamo-depth_amd")
from networks.resnet_encoder import ResnetEncoder
from networks.pose_decoder import PoseDecoder
torch.backends.cudnn.benchmark = True
enc = ResnetEncoder(18, False, num_input_images=2, inp_disp=False).cuda().to(memory_format=torch.channels_last).train()
dec = PoseDecoder(enc.num_ch_enc, num_input_features=1, num_frames_to_predict_for=2).cuda().to(memory_format=torch.channels_last).train()
def run(B, n):
    xs = [torch.rand(B, 6, 192, 640, device="cuda").contiguous(memory_format=torch.channels_last) for _ in range(n)]
    def step():
        tot = 0
        for x in xs:
            a, t = dec([enc(x)])
            tot = tot + a.sum() + t.sum()
        tot.backward()
    for _ in range(4): step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t = time.perf_counter(); e0.record()
    for _ in range(10): step()
    e1.record(); host = time.perf_counter() - t
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 10, host / 10 * 1e3
for B, n in ((12, 2), (24, 1), (12, 2), (24, 1)):
    g, h = run(B, n)
    print("B=%d x %d passes: GPU %.2f ms  host enqueue %.2f ms" % (B, n, g, h))
