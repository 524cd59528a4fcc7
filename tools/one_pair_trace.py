"""GPU probe: N one-pair forwards back to back under `rocprofv3 --kernel-trace --stats`: kernel-time sum per call vs wall clock per call =
how much of the one-pair latency is the GPU waiting between dependent launches.
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d OUT -o b1 -- python REPO/tools/one_pair_trace.py 50)
  python tools/one_pair_trace.py summarize OUT 55"""
import csv
import glob
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def summarize(out_dir, forwards):
    f = glob.glob(os.path.join(out_dir, '**', '*kernel_stats.csv'), recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    eng = [r for r in rows if 'd3r::' in r['Name']]
    tot = sum(float(r['TotalDurationNs']) for r in eng)
    calls = sum(int(r['Calls']) for r in eng)
    print(f'engine kernels: {calls} launches, {tot / 1e6:.1f} ms over {forwards} forwards = {tot / 1e6 / forwards:.2f} ms kernel time and {calls / forwards:.0f} launches per forward')
    for r in sorted(eng, key=lambda r: -float(r['TotalDurationNs']))[:12]:
        print(f"  {float(r['TotalDurationNs']) / 1e6 / forwards:8.2f} ms/forward {int(r['Calls']) / forwards:6.0f} launches/forward avg {float(r['AverageNs']) / 1e3:7.1f} us  {r['Name'][:100]}")


def main():
    if len(sys.argv) > 1 and sys.argv[1] == 'summarize':
        return summarize(sys.argv[2], int(sys.argv[3]))
    import torch
    from bench import build_model, H, W
    from dust3r_amd.synthetic import synthetic_views
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    dev = torch.device('cuda', 0)
    model = build_model('fp16x3', dev)
    v1, v2 = synthetic_views(1, H, W, seed=0, device=dev)
    for _ in range(5):
        model(v1, v2)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        model(v1, v2)
    torch.cuda.synchronize()
    print(f'one pair per call: {(time.perf_counter() - t) / n * 1e3:.2f} ms wall clock per call over {n} calls (+ 5 warm-up calls in the trace)')


if __name__ == '__main__':
    main()
