#!/usr/bin/env python3
"""bench.py -- the driver's measurement contract for the DUSt3R hot path on MI355X.

Metric (BASELINE.json): image-pairs/s of `AsymmetricCroCo3DStereo.forward`, DUSt3R_ViTLarge_BaseDecoder_512_dpt,
synthetic 512x384 pairs (configs[1]: 32 pairs per GPU per step, inputs resident in HBM), at 1/2/4/8 GPUs; the
second half of the metric, global_aligner iterations/s (configs[3]: 20 views, 190 edges, 300 cosine iterations),
is reported in the same JSON line under "aligner".

One step = one engine forward over `--pairs` image pairs per rank. At N > 1 ranks the pairs shard across ranks
(weak scaling: fixed pairs per GPU) and every step ends with the path's ONE collective, the all-gather of the
pairwise predictions (dust3r_amd/parallel.py), issued asynchronously so that it overlaps the next step's compute.

Usage: python bench.py --gpus N --steps K --warmup W [--workload c2|c3|c5]   (N > 1: launched by torch.distributed.run, one rank per GPU)
  --workload c2 (default): BASELINE configs[1], the configuration the metric is quoted on -- what the paragraph above describes.
  --workload c3: configs[2], 20 synthetic views -> make_pairs('complete', symmetrize=False) = 190 pairs, contiguous shards of
                 ceil(190 / N) pairs per rank, every distinct image of a shard encoded once, ONE all-gather per step; a step = the
                 whole 190-pair job (strong scaling: the job is fixed, value = 190 x steps / time).
  --workload c5: configs[4], 100 views, make_pairs('swin-3', symmetrize=True) = 600 pairs: sharded forward + all-gather, then on rank 0
                 global_aligner(PointCloudOptimizer) + compute_global_alignment(init='mst', niter=300, cosine): wall clock per job,
                 next to a CPU figure EXTRAPOLATED by the BASELINE.md section 2 protocol (labelled as such).
Every line carries a "parity_check" block: outputs of the timed 32-pair batch against one-pair-per-call runs of the same pairs
(bit-equality; the one-pair call shape is what tests/ hold to the CPU oracle) and, at N = 1 with the CPU baseline on, the engine
loaded with the CPU oracle's weights against the oracle's own output on the same pair (per-pixel relative error, bar 1e-3).
Prints ONE JSON line on rank 0. Extra objects: "roofline" (dominant kernel, live HIP-event timing), "cpu_baseline"
(the CPU oracle timed on this host's cores on a bounded sample, rank 0 at N=1 only), "aligner", "kernels", "fast_mode"
(the opt-in modes fp16f8 / bf16 / fp16: throughput, their own dominant-kernel roofline block and their measured error against the headline
engine. The headline precision is fp16x3, the engine default and the mode that meets the 1e-3 per-pixel bar on every plain weight seed tested; DESIGN.md section 2 for the outlier-weight figures).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

MODEL = 'DUSt3R_ViTLarge_BaseDecoder_512_dpt'
H, W = 384, 512
GFLOP_PER_PAIR = 1856.8          # 2*MAC over every GEMM / conv / attention contraction of one pair
PEAK_BF16_TFLOPS = 2500.0        # MI355X dense bf16/fp16 MFMA peak (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0            # MI355X HBM3E peak (MI355X_MICROARCH.md)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


STAGE = ['start']        # what this rank is doing, for the error line of guarded_main


class Telemetry:
    """Shader clock / socket power / temperature of THIS rank's GPU, sampled from the amdgpu hwmon files (/sys/class/drm/card*/device/hwmon/hwmon*:
    freq1_input Hz, power1_input uW, power1_cap uW, temp*_input m degC) by a host thread every `period` seconds while a timed region runs. The chip is
    power-managed: two boxes -- or two visits of one box -- run the MFMA-dense kernels at different clocks, and without these numbers beside `value` nobody can
    tell a slower box from a regression (the round-5 driver box ran the nn.Linear launches 9.6 % slower than the builder's with convolutions equal).
    The device's hwmon directory is found through its PCI address; failing that, the card that draws the most power while sampling."""

    def __init__(self, device, period=0.02):
        import glob
        self.period, self.samples, self._stop, self._thread = period, [], None, None
        want = None
        try:
            pr = torch.cuda.get_device_properties(device)
            want = f'{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0'
        except Exception:        # noqa: BLE001 -- older torch: no PCI ids
            pass
        cands = []
        for d in sorted(glob.glob('/sys/class/drm/card*/device/hwmon/hwmon*')):
            if os.path.exists(os.path.join(d, 'power1_input')) and os.path.exists(os.path.join(d, 'freq1_input')):
                cands.append((os.path.basename(os.path.realpath(os.path.join(d, 'device'))), d))
        self.matched = [d for a, d in cands if want and a.lower() == want.lower()]
        self.dirs = self.matched or [d for _, d in cands]
        self.pci = want

    @staticmethod
    def _read(path):
        try:
            with open(path) as f:
                return float(f.read().strip())
        except (OSError, ValueError):
            return None

    def _loop(self):
        while not self._stop.is_set():
            row = []
            for d in self.dirs:
                row.append((self._read(os.path.join(d, 'freq1_input')), self._read(os.path.join(d, 'power1_input')), self._read(os.path.join(d, 'temp2_input'))))
            self.samples.append(row)
            self._stop.wait(self.period)

    def start(self):
        import threading
        if not self.dirs:
            return self
        self.samples, self._stop = [], threading.Event()
        self._thread = threading.Thread(target=self._loop, daemon=True)
        self._thread.start()
        return self

    def stop(self):
        """-> dict for the JSON line (None when the box exposes no hwmon files)."""
        if self._thread is None:
            return None
        self._stop.set()
        self._thread.join()
        if not self.samples:
            return None
        n = len(self.dirs)
        pw = [[r[k][1] for r in self.samples if r[k][1] is not None] for k in range(n)]
        k = max(range(n), key=lambda i: sum(pw[i]) / max(len(pw[i]), 1)) if not self.matched else 0     # unmatched: the card under load
        f = [r[k][0] / 1e6 for r in self.samples if r[k][0] is not None]
        p = [r[k][1] / 1e6 for r in self.samples if r[k][1] is not None]
        t = [r[k][2] / 1e3 for r in self.samples if r[k][2] is not None]
        cap = self._read(os.path.join(self.dirs[k], 'power1_cap'))
        mean = lambda x: sum(x) / len(x) if x else None       # noqa: E731
        return dict(source=self.dirs[k], matched_by_pci_address=bool(self.matched), pci_address=self.pci, samples=len(self.samples), period_s=self.period,
                    sclk_mhz_mean=mean(f), sclk_mhz_min=min(f) if f else None, sclk_mhz_max=max(f) if f else None,
                    power_w_mean=mean(p), power_w_max=max(p) if p else None, power_cap_w=cap / 1e6 if cap else None, temp_c_max=max(t) if t else None)


def nccl_log_path(rank):
    return os.path.join(os.environ.get('TMPDIR', '/tmp'), f'd3r_bench_nccl_{os.environ.get("MASTER_PORT", "0")}_{rank}.log')


def nccl_log_tail(rank, n=12):
    """Last lines RCCL wrote for this rank (NCCL_DEBUG=WARN goes to a per-rank file: main() sets NCCL_DEBUG_FILE) -- warnings only, usually empty."""
    try:
        with open(nccl_log_path(rank)) as f:
            return [ln.rstrip() for ln in f.readlines()[-n:]]
    except OSError:
        return []


def emit(result):
    """ONE strictly valid JSON line: json.dumps would print NaN / Infinity for non-finite floats, which no JSON parser has to accept."""
    def clean(o):
        if isinstance(o, float) and (o != o or o in (float('inf'), float('-inf'))):
            return None
        if isinstance(o, dict):
            return {k: clean(v) for k, v in o.items()}
        if isinstance(o, (list, tuple)):
            return [clean(v) for v in o]
        return o
    print(json.dumps(clean(result), allow_nan=False), flush=True)


def build_model(precision, device):
    from dust3r_amd.model import AsymmetricCroCo3DStereo
    from dust3r_amd.synthetic import MODEL_CONFIGS, OUT_GAIN, synthetic_state_dict
    m = AsymmetricCroCo3DStereo(precision=precision, landscape_only=False, **MODEL_CONFIGS[MODEL])
    t = time.time()
    # weights are generated in HBM and packed by the engine's device kernels: no host copy of the 0.65 G parameters
    m.load_state_dict(synthetic_state_dict({k: torch.empty(v, device='meta') for k, v in m._spec.items()}, 0, OUT_GAIN[MODEL], device=device))
    t1 = time.time()
    m.to(device)
    log(f'[bench] synthetic weights {t1 - t:.1f}s, pack+upload {time.time() - t1:.1f}s, device bytes {m.device_bytes() / 2**30:.2f} GiB')
    return m


GEMM_CFG_NAMES = {0: 'GemmCfg<2,2,4,4,2,128,2> (128x128 tile)', 1: 'GemmCfg<2,4,8,4,2,128,2> (256x256 tile)', 2: 'GemmCfg<2,4,4,4,2,128,2> (256x128 tile)',
                  3: 'GemmCfg<1,8,8,4,2,128,2> (512x128 tile)', 4: 'GemmCfg<1,4,8,4,2,64,3> (256x128, 4 waves)', 5: 'GemmCfg<2,4,8,4,2,64,4> (256x256, 4 stages)', 6: 'GemmCfg<2,4,8,4,2,64,4,1> (256x256, ping-pong)',
                  7: 'GemmCfg<1,4,8,4,2,128,2,6> (256x128 tile, 4 waves, weights of a K step in registers, 2 blocks / CU)',
                  8: 'GemmCfg<2,2,2,2,4,128,2> (64x64 tile, small-batch forwards)',
                  9: 'GemmCfg<1,8,12,3,2,128,2> (M 384 x N 192 tile: whole rounds of 256 CUs for the 24576-row decoder GEMMs)',
                  10: 'gemm_p4_kernel (persistent, M 256 x N 128 tile by four waves with two accumulator sets: epilogue under the next tile\'s K loop)'}


def read_profile(model):
    """Per kernel symbol: gemm_kernel<dtype, cfg> (linear + implicit-GEMM conv launches together), attention, other."""
    from dust3r_amd._lib import lib

    def rd(kind):
        n, ms, work = C.c_int(), C.c_double(), C.c_double()
        if lib.d3r_model_profile_read(model._engine, kind, C.byref(n), C.byref(ms), C.byref(work)) != 0:
            return None
        return dict(launches=n.value, ms=ms.value, gflop=work.value / 1e9)
    out = {'gemm_cfg': {}, 'gemm_f8_cfg': {}, 'linear': dict(launches=0, ms=0.0, gflop=0.0), 'conv': dict(launches=0, ms=0.0, gflop=0.0)}
    zero = dict(launches=0, ms=0.0, gflop=0.0)
    for cfg in range(11):
        lin, cv, f8 = ((rd(cfg), rd(8 + cfg), rd(24 + cfg)) if cfg < 8 else (rd(18), rd(19), dict(zero)) if cfg == 8 else (rd(21), dict(zero), dict(zero)) if cfg == 9
                       else (rd(22), dict(zero), dict(zero)))
        if lin is None or cv is None or f8 is None:
            return None
        for k in ('launches', 'ms', 'gflop'):
            out['linear'][k] += lin[k] + f8[k]
            out['conv'][k] += cv[k]
        if lin['launches'] + cv['launches']:
            out['gemm_cfg'][cfg] = {k: lin[k] + cv[k] for k in ('launches', 'ms', 'gflop')}
        if f8['launches']:          # gemm_kernel<fp16f8, cfg>: a kernel symbol of its own (the transformer blocks' linears of an fp16f8 engine)
            out['gemm_f8_cfg'][cfg] = f8
    out['attention'], out['other'] = rd(16), rd(17)
    return out


def read_launch_table(model):
    """Every launch of the last profiled forward, aggregated by (kernel class, tile configuration, M, N, K): launches, total ms, TFLOP/s.
    This is what tells WHICH GEMM shapes of the network run below the dominant kernel's average."""
    from dust3r_amd._lib import lib
    agg, idx = {}, 0
    kind, M, N, K, ms, work = C.c_int(), C.c_int(), C.c_int(), C.c_int(), C.c_double(), C.c_double()
    while lib.d3r_model_profile_launch(model._engine, idx, C.byref(kind), C.byref(M), C.byref(N), C.byref(K), C.byref(ms), C.byref(work)) == 0:
        idx += 1
        k = kind.value
        name = (f'linear cfg{k}' if k < 8 else f'conv cfg{k - 8}' if k < 16 else 'attention' if k == 16 else f'lin-f8 cfg{k - 24}' if k >= 24
                else 'linear cfg8' if k == 18 else 'conv cfg8' if k == 19 else 'linear cfg9' if k == 21 else 'linear p4' if k == 22 else 'other')
        a = agg.setdefault((name, M.value, N.value, K.value), [0, 0.0, 0.0])
        a[0] += 1
        a[1] += ms.value
        a[2] += work.value
    rows = [dict(kernel=k[0], M=k[1], N=k[2], K=k[3], launches=v[0], ms=round(v[1], 3), tflops=round(v[2] / 1e9 / v[1], 1) if v[1] > 0 else 0.0)
            for k, v in agg.items()]
    return sorted(rows, key=lambda r: -r['ms'])


def pmc_digest(precision=None):
    """The committed digest of the rocprofv3 PMC passes for this precision: profiles/pmc_latest.json (the digest of the LAST visit that ran
    the counter passes, copied there by tools/visit.sh) when it was taken in that precision, else profiles/pmc_<precision>.json (written by
    tools/summarize_prof.py from separate --pmc FETCH_SIZE / --pmc WRITE_SIZE runs of this same command; FETCH_SIZE doubled as
    MI355X_MICROARCH.md prescribes for gfx950). Returns (dict, repo-relative path) or (None, None)."""
    for name in ['pmc_latest.json'] + ([f'pmc_{precision}.json'] if precision else []):
        try:
            with open(os.path.join(ROOT, 'profiles', name)) as f:
                d = json.load(f)
        except Exception:
            continue
        if precision is None or d.get('precision', 'bf16') == precision:
            return d, 'profiles/' + name
    return None, None


def pmc_traffic_gb(cfg, precision=None):
    """(HBM GB per launch of gemm_kernel<precision, cfg> from the committed PMC digest, where it was read from). The counter passes are NOT
    re-collected by this command (they need rocprofv3 around it): the figure is carried from the profile named in `source`."""
    d, path = pmc_digest(precision)
    if d is None:
        return None, None
    v = d.get('gemm_cfg', {}).get(str(cfg), {}).get('hbm_gb_per_launch')
    return v, (f"{path} ({d.get('visit', 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py --single-stream')})" if v is not None else None)


def profile_mode(model, v1, v2, precision, quiet=False):
    """One extra forward with a HIP event in front of every launch (single-stream schedule) -> the roofline block of the dominant kernel,
    the per-kernel-class table and the per-shape launch table of the engine's CURRENT precision mode."""
    from dust3r_amd._lib import lib
    lib.d3r_model_set_option(model._engine, 1, 1)
    model(v1, v2)
    torch.cuda.synchronize()
    prof = read_profile(model)
    table = read_launch_table(model)
    lib.d3r_model_set_option(model._engine, 1, 0)
    if not prof:
        return None
    if not quiet:
        for r in table[:40]:
            log(f"[bench]   {r['kernel']:12s} M={r['M']:8d} N={r['N']:5d} K={r['K']:5d}  x{r['launches']:3d}  {r['ms']:8.3f} ms  {r['tflops']:7.1f} TF/s")
    # dominant kernel = the gemm_kernel instantiation with the largest total time (one kernel symbol in rocprofv3)
    f8name = precision if precision in ('fp16f8', 'fp16x2f8') else 'fp16f8'
    cands = [('fp16x3' if precision in ('fp16f8', 'fp16x2f8') else precision, c, v) for c, v in prof['gemm_cfg'].items()]
    cands += [(f8name, c, v) for c, v in prof['gemm_f8_cfg'].items()]
    dom_dt, dom, d = max(cands, key=lambda t: t[2]['ms'])
    total_ms = prof['linear']['ms'] + prof['conv']['ms'] + prof['attention']['ms'] + prof['other']['ms']
    ach = d['gflop'] / d['ms']                              # GFLOP / ms == TFLOP/s (algorithmic: 2 M N K per launch)
    # MFMA work per logical product in units of one 16-bit MFMA: split-fp16 issues three f16 MFMAs; fp16 + fp8 one f16 MFMA plus
    # both cross terms on the fp8 pipe at twice the 16-bit rate (2 x 2 M N K flops at 5 PFLOP/s = one more 16-bit unit)
    mfma_per_product = {'fp16x3': 3, 'fp16f8': 2, 'fp16x2f8': 2.5}.get(dom_dt, 1)
    cfg_name = GEMM_CFG_NAMES.get(dom, dom)
    if dom_dt in ('fp16f8', 'fp16x2f8') and dom in (1, 2, 3):      # rocprofv3 symbol: gemm_kernel<4 | 5, GemmCfg<...,128,2,4>> (DMA pieces interleaved with the MFMA rows; 4 = fp16f8, 5 = fp16x2f8)
        cfg_name = cfg_name.replace(',128,2>', ',128,2,4>')
    traffic, tsrc = pmc_traffic_gb(dom, dom_dt)
    out = {'roofline': {
        'bound': 'mfma', 'kernel': f'd3r::gemm_kernel<{dom_dt}, {cfg_name}>',
        'achieved': ach, 'peak': PEAK_BF16_TFLOPS, 'unit': 'TFLOP/s', 'frac': ach / PEAK_BF16_TFLOPS,
        'achieved_executed_mfma': ach * mfma_per_product, 'frac_executed_mfma': ach * mfma_per_product / PEAK_BF16_TFLOPS,
        'mfma_per_product': mfma_per_product,
        'traffic': traffic, 'traffic_unit': 'GB per launch (HBM: 2 x FETCH_SIZE + WRITE_SIZE)', 'traffic_source': tsrc,
        'launches_per_step': d['launches'], 'avg_launch_ms': d['ms'] / d['launches'],
        'gflop_per_launch': d['gflop'] / d['launches'], 'share_of_step_time': d['ms'] / total_ms,
        'note': 'achieved / frac = ALGORITHMIC flops (2 M N K per launch, SURVEY 8(d)) over the live HIP-event launch time, against the dense '
                '16-bit MFMA peak; achieved_executed_mfma counts the MFMA time the mode actually issues in 16-bit-MFMA units per product '
                '(fp16x3: 3 f16 MFMAs; fp16f8: 1 f16 MFMA + both cross terms on one e4m3 MFMA at twice the rate = 2 units); traffic is NOT '
                're-measured by this command: it is carried from the committed rocprofv3 PMC digest named in traffic_source',
        'timing': 'HIP events around every launch on the launch stream, single-stream schedule, one extra forward after the timed region'}}
    # every GEMM-shaped launch of the step together (nn.Linear + implicit-GEMM convolutions, all tile configurations incl. the persistent kernel): since round 6 the
    # fastest nn.Linear launches run on another kernel symbol (cfg10), so the dominant symbol alone no longer describes "the GEMMs"
    g_ms = prof['linear']['ms'] + prof['conv']['ms']
    g_gf = prof['linear']['gflop'] + prof['conv']['gflop']
    out['roofline']['all_gemm_launches'] = {'achieved': g_gf / g_ms, 'frac': g_gf / g_ms / PEAK_BF16_TFLOPS, 'unit': 'TFLOP/s', 'ms_per_step': g_ms,
                                            'share_of_step_time': g_ms / total_ms, 'launches_per_step': prof['linear']['launches'] + prof['conv']['launches']}
    kern = {f'gemm_kernel cfg{c}': dict(v, tflops=v['gflop'] / v['ms']) for c, v in prof['gemm_cfg'].items()}
    kern.update({f'gemm_kernel<fp16f8> cfg{c}': dict(v, tflops=v['gflop'] / v['ms']) for c, v in prof['gemm_f8_cfg'].items()})
    for k in ('attention', 'other'):
        v = prof[k]
        kern[k] = dict(v, tflops=(v['gflop'] / v['ms'] if v['ms'] > 0 else 0.0))
    kern['all_gemm_linear'] = dict(prof['linear'], tflops=prof['linear']['gflop'] / max(prof['linear']['ms'], 1e-9))
    kern['all_gemm_conv'] = dict(prof['conv'], tflops=prof['conv']['gflop'] / max(prof['conv']['ms'], 1e-9))
    out['kernels'] = kern
    out['launch_table'] = table[:24]
    if not quiet:
        log('[bench] per-kernel: ' + ', '.join(f"{k} {v['ms']:.1f} ms / {v['launches']} launches / {v['tflops']:.0f} TF/s" for k, v in kern.items()))
    return out


def rel_stats(got, ref):
    """per pixel ||got - ref||_2 / max(||ref||_2, 1e-8) -> max / p99.99 / p99 / mean."""
    e = ((got.float().cpu() - ref.float().cpu()).norm(dim=-1) / ref.float().cpu().norm(dim=-1).clamp_min(1e-8)).flatten()
    srt = e.sort().values
    q = lambda f: float(srt[min(int(f * srt.numel()), srt.numel() - 1)])   # noqa: E731
    return {'max': float(srt[-1]), 'p99.99': q(0.9999), 'p99': q(0.99), 'mean': float(e.mean())}


def parity_batch_vs_single(model, v1, v2, full, picks):
    """The timed batch's outputs (`full` = pts1, conf1, pts2, conf2 of the whole batch) against one-pair-per-call runs of pairs `picks`:
    the call shape tests/test_forward_gpu.py::test_full_size_fp32_pair_matches_oracle holds to the CPU oracle. Bit-equality expected:
    every kernel is batch-position independent and no tile choice changes the order of a K sum."""
    worst, equal, sk_rel = 0.0, True, 0.0
    for b in picks:
        s1 = dict(img=v1['img'][b:b + 1], true_shape=v1['true_shape'][b:b + 1], idx=[0], instance=['0'])
        s2 = dict(img=v2['img'][b:b + 1], true_shape=v2['true_shape'][b:b + 1], idx=[1], instance=['1'])
        # (a) every K sum in one block (the default): the one-pair call must be BIT-equal to its place in the batch
        o1, o2 = model(s1, s2)
        for a, w in zip((o1['pts3d'], o1['conf'], o2['pts3d_in_other_view'], o2['conf']), full):
            equal = equal and bool(torch.equal(a[0], w[b]))
            worst = max(worst, float((a[0] - w[b]).abs().max()))
        # (b) the opt-in split-K one-pair call (D3R_MODEL_OPT_SPLIT_K: a different summation order of K in its small launches): fp32-rounding distance
        model.set_split_k(True)
        o1, o2 = model(s1, s2)
        model.set_split_k(False)
        for a, w in ((o1['pts3d'][0], full[0][b]), (o2['pts3d_in_other_view'][0], full[2][b])):
            sk_rel = max(sk_rel, float(((a - w).norm(dim=-1) / w.norm(dim=-1).clamp_min(1e-8)).max()))
    finite = all(bool(torch.isfinite(t).all()) for t in full)
    return {'what': f'pairs {list(picks)} of the timed {full[0].shape[0]}-pair batch vs the same pairs run one per call (forward outputs pts3d / conf of both views): '
                    'bit-equal (default engine); per-pixel relative distance of the opt-in split-K one-pair call',
            'bit_equal': equal, 'max_abs_diff': worst, 'split_k_one_pair_call_max_rel_diff': sk_rel, 'all_outputs_finite': finite, 'pass': equal and finite and sk_rel < 2e-4}


def parity_vs_cpu_oracle(model, oracle, v1, v2, ref):
    """Engine (loaded with the CPU oracle's weights) vs the oracle's own fp32 output on the same 512x384 pair: the north-star measure
    (per-pixel relative pointmap error, bar 1e-3). `ref` = (r1, r2) from the oracle call bench.py's cpu_baseline leg makes anyway."""
    model.load_state_dict(oracle.state_dict(), strict=True)
    e1, e2 = model(v1, v2)
    st = rel_stats(torch.cat((e1['pts3d'], e2['pts3d_in_other_view'])), torch.cat((ref[0]['pts3d'], ref[1]['pts3d_in_other_view'])))
    cf = float(((torch.cat((e1['conf'], e2['conf'])).cpu() - torch.cat((ref[0]['conf'], ref[1]['conf']))).abs() / torch.cat((ref[0]['conf'], ref[1]['conf']))).max())
    return {'what': f'{MODEL} one 512x384 pair, engine precision {model.precision} with the CPU oracle\'s weights vs oracle/dust3r_ref.py fp32 on this host',
            'pointmap_rel_err': st, 'conf_rel_err_max': cf, 'tolerance': 1e-3, 'pass': st['max'] < 1e-3 and cf < 3e-3}


def bench_aligner(device, niter=300, n_views=20, symmetrize=False):
    """configs[3]: PointCloudOptimizer, 20 synthetic views -> 190 edges (symmetrize=True: the demo's symmetrised graph, dust3r/demo.py:155, 380 edges),
    300 iterations, cosine schedule, one GPU."""
    from dust3r_amd.cloud_opt import global_aligner
    from dust3r_amd.cloud_opt.base_opt import global_alignment_loop
    from dust3r_amd.synthetic import synthetic_scene
    out, init, gt = synthetic_scene(n_views, H, W, seed=0, symmetrize=symmetrize, device=device, device_rng=symmetrize)
    scene = global_aligner(out, device, verbose=False)
    scene.load_state_dict(init)
    E, n, A = scene.n_edges, scene.n_imgs, H * W
    global_alignment_loop(scene, niter=5)                      # warm-up (also builds the engine)
    runs = []
    for _ in range(3):          # the whole 300-iteration run three times from the same state (75 ms each): MEDIAN reported, every run listed
        scene.load_state_dict(init)       # (900 dependent launches: one run of the round-5 visit H read 128 ms on a box whose other runs read 74)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        loss = global_alignment_loop(scene, niter=niter, schedule='cosine', lr=0.01)
        e1.record()
        torch.cuda.synchronize()
        runs.append(e0.elapsed_time(e1))
    ms = sorted(runs)[1]
    bytes_iter = E * A * 32 + n * A * 4 * 6                    # preds + weights once, depth param/Adam r/w
    gbs = bytes_iter * niter / (ms * 1e-3) / 1e9
    traffic, tsrc = None, None
    for prec in ('fp16x3', None):
        d, path = pmc_digest(prec)
        if d and d.get('aligner_main_kernel', {}).get('hbm_gb_per_launch') is not None:
            traffic, tsrc = d['aligner_main_kernel']['hbm_gb_per_launch'], path
            break
    res = dict(metric='global_aligner_iters_per_sec', value=niter / (ms * 1e-3), unit='iters/s', n_views=n, n_edges=E, niter=niter,
               ms_total=ms, ms_runs=runs, final_loss=loss,
               roofline=dict(bound='hbm', kernel='d3r::aligner_main_kernel (+ reduce + small kernels: whole iteration timed)', achieved=gbs,
                             peak=PEAK_HBM_GBS, unit='GB/s', frac=gbs / PEAK_HBM_GBS, bytes_per_iter=bytes_iter, traffic=traffic, traffic_source=tsrc))
    return res, (out, init)


def _keep_heap():
    """glibc: serve big blocks from the heap and never trim it, so that page faults are paid once (the sandboxed hosts
    fault slowly); affects only the CPU-baseline legs."""
    try:
        libc = C.CDLL('libc.so.6')
        libc.mallopt(-3, 1 << 30)   # M_MMAP_THRESHOLD
        libc.mallopt(-1, 1 << 40)   # M_TRIM_THRESHOLD
    except Exception:
        pass


def _cpu_model():
    try:
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.startswith('model name'):
                    return line.split(':', 1)[1].strip()
    except Exception:
        pass
    return 'unknown'


def cpu_baseline_forward(reps=3, keep=None):
    """SURVEY.md 8(d) protocol: the CPU oracle (fp32 PyTorch restatement of the reference path, oracle/dust3r_ref.py) on this
    host's cores, same model / image size as the GPU leg, B = 1 and B = 4 pairs per call, `reps` calls each after one untimed
    call, MEDIAN reported. `value` is the better of the two medians in pairs/s (both are in `sample`)."""
    from oracle.dust3r_ref import build_ref_model_fast
    from dust3r_amd.synthetic import synthetic_views
    from oracle import tune_threads
    tune_threads()
    _keep_heap()
    t = time.time()
    oracle = build_ref_model_fast(MODEL)
    log(f'[bench] cpu oracle built in {time.time() - t:.1f}s, threads {torch.get_num_threads()}')
    med = {}
    with torch.no_grad():
        for B in (1, 4):
            v1, v2 = synthetic_views(B, H, W, seed=0)
            r = oracle(v1, v2)                  # untimed: first-touch page faults
            if B == 1 and keep is not None:     # the parity_check block compares the engine with this output
                keep.update(oracle=oracle, views=(v1, v2), ref=r)
            times = []
            for _ in range(reps):
                t = time.time()
                oracle(v1, v2)
                times.append(time.time() - t)
            med[B] = sorted(times)[len(times) // 2]
            log(f'[bench] cpu oracle B={B}: median {med[B]:.2f} s per call ({B / med[B]:.3f} pairs/s)')
    best = max(B / med[B] for B in med)
    return dict(value=best, unit='pairs/s', cores=torch.get_num_threads(), cpu_model=_cpu_model(), logical_cpus=os.cpu_count(), kind='port',
                pairs_per_s_B1=1 / med[1], pairs_per_s_B4=4 / med[4],
                sample=f'{MODEL} fp32 512x384 (oracle/dust3r_ref.py), median of {reps} calls after 1 untimed: B=1 {med[1]:.2f} s/call, B=4 {med[4]:.2f} s/call')


def cpu_baseline_aligner(scene_io, warm=2, timed=20):
    """SURVEY.md 8(d) protocol: the CPU oracle of the aligner loop (oracle/aligner_ref.py: the reference's forward restated +
    torch autograd + Adam) on the SAME 20-view / 190-edge 512x384 inputs and initial state as the GPU leg: `warm` untimed
    iterations, then `timed` iterations on the clock (no edge-count extrapolation)."""
    from oracle.aligner_ref import AlignerRef
    out, init = scene_io
    out = {k: ({kk: (vv.cpu() if isinstance(vv, torch.Tensor) else vv) for kk, vv in v.items()} if isinstance(v, dict) else v)
           for k, v in out.items()}
    ref = AlignerRef(out).load_state(init)
    E = len(ref.edges)
    ref.run(niter=warm, total=300)
    t = time.time()
    ref.run(niter=timed, total=300, start=warm)
    dt = time.time() - t
    return dict(value=timed / dt, unit='iters/s', cores=torch.get_num_threads(), cpu_model=_cpu_model(), kind='port',
                sample=f'{timed} timed iterations (after {warm} untimed) of the same {ref.n_imgs}-view / {E}-edge 512x384 scene and initial state as the GPU leg: '
                       f'{dt:.1f} s = {dt / timed:.3f} s/iter; 300 iterations extrapolate to {300 * dt / timed:.0f} s')


ENC_GFLOP_PER_IMAGE = 523.0                  # encoder incl. patch embedding, per 512x384 image
DEC_HEAD_GFLOP_PER_PAIR = 2 * 218.6 + 2 * 186.7   # both decoders (incl. decoder_embed) + both DPT heads, per pair


def run_sharded(args, model, world, rank, device, one_device, backend):
    """--workload c3 / c5 (BASELINE configs[2] / configs[4]): a FIXED job (190 / 600 pairs over 20 / 100 views) cut into contiguous
    shards of ceil(P / N) pairs, dust3r_amd.parallel's per-rank routine (every distinct image of the shard through the encoder once,
    the heads write the packed (pairs, H, W, 8) payload in place), ONE all-gather; c5 then aligns the gathered predictions on rank 0
    (global_aligner + init='mst' + 300 cosine iterations, dust3r/demo.py:158-178). Images are resident in HBM before the timed region."""
    from dust3r_amd.image_pairs import make_pairs
    from dust3r_amd.parallel import _local_same_size, all_gather_packed, shard_plan, unpack_predictions
    from dust3r_amd.synthetic import synthetic_image_list
    c5 = args.workload == 'c5'
    n_views, graph, sym = (100, 'swin-3', True) if c5 else (20, 'complete', False)
    imgs = synthetic_image_list(n_views, H, W, seed=0)
    for v in imgs:
        v['img'] = v['img'].to(device)
    pairs = make_pairs(imgs, scene_graph=graph, prefilter=None, symmetrize=sym)
    P = len(pairs)
    plan = shard_plan(pairs, world, encode_once=True)        # dust3r_amd.parallel: which pairs each rank runs (cost-balanced, same on every rank)
    per, counts, images_per_rank = plan.per, plan.counts, plan.images
    my_pairs = [pairs[k] for k in plan.shard(rank)]
    log(f'[bench] rank {rank}: {args.workload}: {P} pairs over {n_views} views, shard of {len(my_pairs)} pairs touching {images_per_rank[rank]} distinct images ({plan.name})')
    keep = plan.source.to(device)
    gathered = torch.empty((world * per, H, W, 8), dtype=torch.float32, device=device) if world > 1 else None
    scene_payload = scene_views = scene_gt = None
    if c5 and rank == 0:
        # Stage B's VALUES. Random-init weights do not produce a scene (the pointmaps of the timed forward are finite but geometrically
        # meaningless, and the MST / Procrustes / focal initialisation of such input ends in NaN -- in the unmodified reference too: checked on
        # the tiny model, its Weiszfeld focal is 0 and log(0) follows). So the values the aligner sees are those of a geometrically consistent
        # synthetic scene OF THE SAME SHAPE (same 100 views, same 600 edges in make_pairs' order, 512x384), packed ahead of the timed region
        # in the engine heads' payload format -- and they reach global_aligner THROUGH stage A's hand-over path (round 5): the gathered
        # (world x per, H, W, 8) payload -> index_select(plan.source) -> the substitution of the values -> unpack_predictions -> the
        # dict inference() returns -> global_aligner. Stage B consumes exactly the tensors stage A's code hands over; only their contents differ.
        from dust3r_amd.parallel import pack_predictions
        from dust3r_amd.synthetic import synthetic_scene
        t = time.time()
        scene_out, _, scene_gt = synthetic_scene(n_views, H, W, seed=0, scene_graph=graph, symmetrize=sym, noise=0.002, device=device, device_rng=True)
        assert scene_out['view1']['idx'] == [int(p[0]['idx']) for p in pairs] and scene_out['view2']['idx'] == [int(p[1]['idx']) for p in pairs]
        scene_payload = pack_predictions(scene_out['pred1'], scene_out['pred2'])
        scene_views = (scene_out['view1'], scene_out['view2'])
        del scene_out
        log(f'[bench] consistent synthetic scene for the alignment stage built + packed in {time.time() - t:.1f} s')
    if c5 and world > 1:
        # Stage B on ALL ranks (round 5: compute_global_alignment(group=...), one contiguous range of images per rank, one all-reduce of the reduced sums per iteration,
        # bit-identical to the one-GPU loop): every rank needs the scene's values and its view metadata -- broadcast once, ahead of the timed region
        if rank != 0:
            scene_payload = torch.empty((P, H, W, 8), dtype=torch.float32, device=device)
        dist.broadcast(scene_payload, src=0)
        meta = [scene_views if rank == 0 else None]
        dist.broadcast_object_list(meta, src=0)
        scene_views = meta[0]
    stage = {}

    ev_g0, ev_g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def step():
        STAGE[0] = f'{args.workload}: forward of the shard'
        local = _local_same_size(my_pairs, per, model, device, args.pairs, device, True, H, W)
        if world > 1:
            STAGE[0] = f'{args.workload}: all_gather of the packed predictions ({world} x {per} pairs x {H * W * 32 / 1e6:.1f} MB)'
            ev_g0.record()
            all_gather_packed(local, out=gathered)
            ev_g1.record()
            stage['gather_events'] = True
            allp = gathered
        else:
            allp = local
        if not c5:
            return allp, None
        from dust3r_amd.cloud_opt import GlobalAlignerMode, global_aligner
        STAGE[0] = 'c5: global_aligner + init=mst + 300 iterations'
        torch.cuda.synchronize()
        t = time.perf_counter()
        handed = allp.index_select(0, keep)                       # the caller's pair order (drops padding rows, undoes the shard plan's order)
        stage['gathered_finite'] = bool(torch.isfinite(handed).all())
        handed.copy_(scene_payload)                               # same tensor, same layout: the consistent scene's values (see above)
        pred1, pred2 = unpack_predictions(handed)                 # the hand-over format of inference() / inference_sharded()
        del handed
        output = dict(view1=scene_views[0], view2=scene_views[1], pred1=pred1, pred2=pred2, loss=None)
        scene = global_aligner(output, device, mode=GlobalAlignerMode.PointCloudOptimizer, verbose=False)
        loss = scene.compute_global_alignment(init='mst', niter=300, schedule='cosine', lr=0.01, group=True if world > 1 else None)
        poses, focals = scene.get_im_poses(), scene.get_focals()
        torch.cuda.synchronize()
        stage['align_s'] = time.perf_counter() - t
        stage['loss'], stage['poses_finite'] = float(loss), bool(torch.isfinite(poses).all())
        if scene_gt is not None:
            stage['focal_err'] = float((focals.detach().flatten().cpu() / scene_gt['focal'] - 1).abs().max())
        return allp, scene

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    last = None
    for _ in range(args.steps):
        last = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    STAGE[0] = 'rank table'
    gather_ms = None
    if world > 1 and stage.get('gather_events'):
        torch.cuda.synchronize()
        g = torch.tensor([ev_g0.elapsed_time(ev_g1)], dtype=torch.float64, device=device)        # the LAST job's collective on this rank (incl. waiting for the slowest rank's shard)
        dist.all_reduce(g, op=dist.ReduceOp.MAX)
        gather_ms = float(g.item())
    who = rank_table(world, rank, device, len(my_pairs))
    if rank != 0:
        return None
    sec = dt / args.steps
    gflop = n_views * ENC_GFLOP_PER_IMAGE + P * DEC_HEAD_GFLOP_PER_PAIR if world == 1 else sum(images_per_rank) * ENC_GFLOP_PER_IMAGE + P * DEC_HEAD_GFLOP_PER_PAIR
    selftest = ' (SELF-TEST: all ranks on one device, backend ' + backend + ' -- not a scaling measurement)' if one_device and world > 1 else ''
    common = dict(n_gpus=world, rccl_world_size=who['rccl_world_size'], collective_backend=who['backend'], distinct_devices=who['distinct_devices'], ranks=who['ranks'], steps=args.steps, warmup=args.warmup, ms_per_step=sec * 1e3, scaling='strong', vs_baseline=None, dtype=args.precision,
                  gather_ms=gather_ms, gather_note='HIP-event time of the ONE all-gather of the last job, max over ranks (includes waiting for the slowest shard); null at N = 1: nothing is gathered',
                  data='synthetic' + selftest)
    cfg = {'workload': f'{MODEL}, {n_views} synthetic 512x384 views -> make_pairs({graph!r}, symmetrize={sym}) = {P} pairs (BASELINE configs[{4 if c5 else 2}]), cost-balanced shards (dust3r_amd.parallel.shard_plan: <= '
                       f'{per} pairs per rank), each distinct image of a shard encoded once, ONE all-gather of the packed predictions per job'
                       + ('; then on every rank global_aligner(PointCloudOptimizer) + init=mst (replicated) + 300 cosine Adam iterations SHARED by the ranks (a contiguous range of images each, one all-reduce of the reduced sums per iteration: compute_global_alignment(group=...)); values of a consistent synthetic scene of the same shape, fed through the gathered payload tensor' if c5 else '') + '; random-init weights, images resident in HBM',
           'pairs': P, 'views': n_views, 'pairs_per_rank': counts, 'distinct_images_per_rank': images_per_rank,
           'distinct_images_per_rank_max_min': [max(images_per_rank), min(images_per_rank)], 'shard_plan': plan.summary(), 'pairs_per_engine_call': args.pairs,
           'parallelism': f'pair-sharded dp{world}'}
    if not c5:
        result = dict(metric='image_pairs_per_sec_forward_512x384_190_pairs_sharded', value=P / sec, unit='pairs/s', higher_is_better=True, config=cfg,
                      forward_gflop_executed_per_job=gflop, forward_tflops_executed=gflop / sec / 1e3,
                      note='encode-once: the job executes fewer flops than pairs x 1856.8 GFLOP (the pair-by-pair schedule of the reference); value counts PAIRS', **common)
    else:
        result = dict(metric='end_to_end_seconds_100_views_forward_plus_global_aligner_on_consistent_scene_of_same_shape', value=sec, unit='s per job', higher_is_better=False, config=cfg,
                      stages=dict(forward_and_gather_s=sec - stage.get('align_s', 0.0), aligner_build_init_300_iters_s=stage.get('align_s'), final_loss=stage.get('loss'),
                                  poses_finite=stage.get('poses_finite'), focal_error_vs_ground_truth=stage.get('focal_err'), gathered_predictions_finite=stage.get('gathered_finite'),
                                  note='stage split from the LAST job; value is the mean over the timed jobs. The alignment stage runs on a geometrically consistent synthetic '
                                       'scene of the same shape (100 views, the same 600 edges, 512x384) whose values are substituted INTO the handed-over payload tensor: gathered payload -> index_select(plan.source) -> '
                                       'values substituted -> unpack_predictions -> global_aligner, i.e. stage B consumes the tensors stage A hands over; random-init weights do not produce a scene'),
                      pairs_per_s_end_to_end=P / sec, forward_gflop_executed_per_job=gflop, **common)
    # parity of the job's own outputs: sampled pairs of the gathered payload vs one-pair-per-call runs (bit-equality)
    if not args.no_parity:
        allp = last[0].index_select(0, keep) if world > 1 else last[0]
        equal, worst = True, 0.0
        picks = sorted({0, P // 3, P // 2, P - 1})
        for k in picks:
            a, b = pairs[k]
            pk = model.forward_packed(dict(img=a['img']), dict(img=b['img']))
            equal = equal and bool(torch.equal(pk[0], allp[k]))
            worst = max(worst, float((pk[0] - allp[k]).abs().max()))
        result['parity_check'] = {'sharded_job_vs_single_pair_calls': {'what': f'pairs {picks} of the gathered {P}-pair payload vs forward_packed of that pair alone', 'bit_equal': equal,
                                                                       'max_abs_diff': worst, 'all_outputs_finite': bool(torch.isfinite(allp).all()), 'pass': equal}}
        log(f"[bench] parity_check: {result['parity_check']}")
    log(f"[bench] {args.workload}: {P} pairs per job, {sec:.3f} s per job on {world} GPU(s)" + (f", aligner stage {stage.get('align_s', 0):.3f} s, loss {stage.get('loss')}" if c5 else ''))
    last = None
    if c5 and world == 1 and not args.no_cpu_baseline:
        try:      # CPU wall clock EXTRAPOLATED from per-pair and per-iteration medians of the oracle on this host
            fw = cpu_baseline_forward()
            al_gpu, scene_io = bench_aligner(device, niter=30)
            al = cpu_baseline_aligner(scene_io)
            cpu_s = P / fw['value'] + 300 * (P / 190.0) / al['value']
            result['cpu_baseline'] = dict(value=cpu_s, unit='s per job (EXTRAPOLATED)', cores=fw['cores'], cpu_model=fw['cpu_model'], kind='port',
                                          sample=f"EXTRAPOLATED, not run: {P} pairs / ({fw['value']:.3f} pairs/s: {fw['sample']}) + 300 iterations x ({P}/190 edges) / ({al['value']:.3f} iters/s at 190 edges: {al['sample']}); "
                                                 'the MST / PnP initialisation of the reference is not in the CPU figure',
                                          gpu_over_cpu=cpu_s / sec)
        except Exception as e:
            result['cpu_baseline'] = {'error': repr(e)}
    return result


def self_launch(args):
    """`python bench.py --gpus N` with no torchrun around it (WORLD_SIZE unset): re-exec this command under torch.distributed.run, one rank
    per GPU of this node, rendezvous on 127.0.0.1 -- the same launch line the driver uses for N > 1. Does not return."""
    import socket
    n_dev = torch.cuda.device_count()
    if os.environ.get('D3R_BENCH_ONE_DEVICE') != '1' and n_dev < args.gpus:
        log(f'[bench] --gpus {args.gpus} but this node exposes {n_dev} GPU(s): one rank per GPU is the contract (D3R_BENCH_ONE_DEVICE=1 + D3R_BENCH_BACKEND=gloo '
            'runs every rank on cuda:0 as a self-test of the code path)')
        sys.exit(2)
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'), D3R_BENCH_SELF_LAUNCHED='1')
    from dust3r_amd.utils.device import usable_cpus
    env.setdefault('OMP_NUM_THREADS', str(max(1, usable_cpus() // max(args.gpus, 1))))       # the CPUs this container may use (not the box's logical CPUs), shared by the ranks
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus), '--master-addr', '127.0.0.1', '--master-port', str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    log('[bench] WORLD_SIZE unset: launching ' + ' '.join(cmd))
    sys.stdout.flush()
    sys.stderr.flush()
    os.execvpe(sys.executable, cmd, env)


def rank_table(world, rank, device, units):
    """What actually ran: one row per rank (device index, device name, PCI bus id when torch exposes it, units of work per step), gathered
    over the process group, and the world size the COLLECTIVE LIBRARY reports (dist.get_world_size(), not the --gpus argument)."""
    prop = torch.cuda.get_device_properties(device)
    row = dict(rank=rank, device=str(device), name=prop.name, gcn_arch=getattr(prop, 'gcnArchName', None), pci_bus_id=getattr(prop, 'pci_bus_id', None),
               uuid=str(getattr(prop, 'uuid', '')) or None, pairs_per_step=units, host=os.uname().nodename, pid=os.getpid(), nccl_warnings=nccl_log_tail(rank))
    if world > 1 and dist.is_initialized():
        rows = [None] * world
        dist.all_gather_object(rows, row)
        return dict(rccl_world_size=dist.get_world_size(), backend=dist.get_backend(), ranks=rows,
                    distinct_devices=len({(r['host'], r['device'], r.get('uuid') or r.get('pci_bus_id')) for r in rows}))
    return dict(rccl_world_size=dist.get_world_size() if dist.is_initialized() else 1, backend=dist.get_backend() if dist.is_initialized() else None,
                ranks=[row], distinct_devices=1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--pairs', type=int, default=32, help='image pairs per GPU per step (configs[1]: 32)')
    ap.add_argument('--precision', default=os.environ.get('DUST3R_AMD_PRECISION', 'fp16x3'),
                    help='engine precision of the HEADLINE: fp16x3 (default, the engine default: meets the 1e-3 per-pixel pointmap bar); fp16x2f8 / fp16f8 / bf16 / fp16 are opt-in modes reported under fast_mode')
    ap.add_argument('--workload', default='c2', choices=['c2', 'c3', 'c5'], help='c2 (default): BASELINE configs[1], 32 pairs per GPU per step; c3: configs[2], 20 views -> 190 pairs sharded; c5: configs[4], 100 views swin-3 -> 600 pairs, forward + global_aligner end to end')
    ap.add_argument('--no-parity', action='store_true', help='skip the parity_check block')
    ap.add_argument('--no-aligner', action='store_true')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-profile', action='store_true')
    ap.add_argument('--no-latency', action='store_true', help='skip the latency block (1 / 2 / 4 / 8 pairs per call)')
    ap.add_argument('--no-aligner-380', action='store_true', help='skip the second aligner scene (the demo\'s symmetrised graph, 380 edges): under rocprofv3 its launches would be averaged into the same kernel symbol as the BASELINE scene\'s')
    ap.add_argument('--no-fast', '--no-accurate', dest='no_fast', action='store_true', help='skip the fast_mode block (fp16f8 / bf16 / fp16 throughput + their measured error)')
    ap.add_argument('--single-stream', action='store_true', help='keep decoder side 2 / head 2 on the main stream (serialised kernels: use under rocprofv3 so that per-kernel durations are not inflated by overlap)')
    args = ap.parse_args()

    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        self_launch(args)                     # bare `python bench.py --gpus N`: becomes N ranks under torch.distributed.run
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    if world != args.gpus:                    # an external launcher decides how many ranks exist; the line reports what ran (n_gpus = WORLD_SIZE)
        log(f'[bench] note: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} rank(s): running and reporting {world}')
    # D3R_BENCH_ONE_DEVICE=1: every rank on cuda:0 (self-test of the N > 1 code path on a one-GPU box; RCCL refuses two ranks on one
    # device, so pair it with D3R_BENCH_BACKEND=gloo -- the collective then stages through the host: NOT a measurement of anything)
    one_device = os.environ.get('D3R_BENCH_ONE_DEVICE') == '1'
    backend = os.environ.get('D3R_BENCH_BACKEND', 'nccl')
    dev_index = 0 if one_device else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device('cuda', dev_index)
    force_gather = os.environ.get('D3R_BENCH_FORCE_GATHER') == '1'     # exercise the collective path on one GPU (self-test)
    if world > 1 or force_gather:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29517')
        import datetime
        os.environ.setdefault('NCCL_DEBUG', 'WARN')                       # RCCL reads the NCCL_* variables; warnings of each rank go to its own file and into the rank table
        os.environ.setdefault('NCCL_DEBUG_FILE', nccl_log_path(rank))
        STAGE[0] = f'init_process_group({backend})'
        tmo = datetime.timedelta(seconds=int(os.environ.get('D3R_BENCH_COLLECTIVE_TIMEOUT', '300')))
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=device, timeout=tmo)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=tmo)
        STAGE[0] = 'first collective'
        probe = torch.ones(1, device=device)
        dist.all_reduce(probe)                                            # the first collective builds the communicators: fail HERE, with a name, not inside the timed loop
        torch.cuda.synchronize()
        assert int(probe.item()) == world, f'all_reduce of ones returned {probe.item()} on a world of {world}'
    STAGE[0] = 'build model'

    from dust3r_amd import _lib
    from dust3r_amd.parallel import all_gather_packed
    from dust3r_amd.synthetic import synthetic_views
    _lib.require_device()
    # DUST3R_CKPT=<released checkpoint file>: the first thing a box with weights on it does is pin the croco / roma restatements of oracle/ against them
    # (tools/validate_checkpoint.py: key-for-key load into oracle and engine, per-pixel statistics of one pair per precision) -- BEFORE anything is timed, in its own
    # process; the verdict travels in the JSON line (`checkpoint_validation`). The timed workload itself stays BASELINE's: synthetic inputs, random-init weights.
    ckpt_validation = None
    if rank == 0 and os.environ.get('DUST3R_CKPT'):
        ckpt_validation = validate_checkpoint_first(os.environ['DUST3R_CKPT'])
    model = build_model(args.precision, device)
    if args.single_stream:
        model.set_two_streams(False)
    B = args.pairs
    if args.workload != 'c2':
        result = run_sharded(args, model, world, rank, device, one_device, backend)
        if rank == 0:
            emit(result)
        if world > 1:
            dist.barrier()
        if dist.is_initialized():
            dist.destroy_process_group()
        return
    v1, v2 = synthetic_views(B, H, W, seed=rank, device=device)      # resident in HBM before the timed region

    do_gather = world > 1 or force_gather
    gather_stream = torch.cuda.Stream(device=device) if do_gather else None
    gather_out = [torch.empty((world * B, H, W, 8), dtype=torch.float32, device=device) for _ in range(2)] if do_gather else None
    pending = []

    def step(i):
        if not do_gather:
            return model(v1, v2)
        packed = model.forward_packed(v1, v2)                # the heads write the (B,H,W,8) all-gather payload directly
        ev = torch.cuda.Event()
        ev.record()
        gather_stream.wait_event(ev)
        with torch.cuda.stream(gather_stream):
            packed.record_stream(gather_stream)
            _, work = all_gather_packed(packed, async_op=True, out=gather_out[i & 1])
        pending.append(work)
        if len(pending) > 1:                                 # at most one all-gather in flight behind the compute
            pending.pop(0).wait()
        return packed

    def drain():
        while pending:
            pending.pop(0).wait()
        if gather_stream is not None:
            torch.cuda.current_stream().wait_stream(gather_stream)

    for i in range(args.warmup):
        step(i)
    drain()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    STAGE[0] = 'c2: timed steps'
    tele = Telemetry(device).start()
    t0 = time.perf_counter()
    last = None
    for i in range(args.steps):
        last = step(i)
    drain()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    telemetry = tele.stop()
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    who = rank_table(world, rank, device, B)

    result = None
    if rank == 0:
        pairs_total = world * B * args.steps
        value = pairs_total / dt
        result = {
            'metric': 'image_pairs_per_sec_forward_512x384', 'value': value, 'unit': 'pairs/s', 'n_gpus': world, 'rccl_world_size': who['rccl_world_size'],
            'collective_backend': who['backend'], 'distinct_devices': who['distinct_devices'], 'ranks': who['ranks'], 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': args.precision, 'data': 'synthetic' + (' (SELF-TEST: all ranks on one device, backend ' + backend + ' -- not a scaling measurement)' if one_device and world > 1 else ''),
            'config': {'workload': f'{MODEL} AsymmetricCroCo3DStereo.forward, {B} synthetic 512x384 pairs per GPU per step (BASELINE configs[1]), '
                                   'random-init weights, inputs resident in HBM' + ('; pairs sharded over ranks + one all-gather of the pairwise predictions per step' if world > 1 else ''),
                       'pairs_per_gpu': B, 'global_pairs_per_step': world * B, 'parallelism': f'pair-sharded dp{world}'},
            'forward_tflops_per_gpu': value / world * GFLOP_PER_PAIR / 1e3,
            'forward_frac_of_bf16_mfma_peak': value / world * GFLOP_PER_PAIR / 1e3 / PEAK_BF16_TFLOPS,
        }
        if ckpt_validation is not None:
            result['checkpoint_validation'] = ckpt_validation
        result['telemetry'] = telemetry
        if telemetry:
            result['sclk_mhz_mean'], result['power_w_mean'] = telemetry['sclk_mhz_mean'], telemetry['power_w_mean']
        log(f'[bench] {value:.2f} pairs/s on {world} GPU(s), {dt / args.steps * 1e3:.1f} ms/step; telemetry {telemetry}')

    # ---- parity of the timed configuration itself: the last timed step's outputs vs one-pair-per-call runs -----------
    if rank == 0 and not args.no_parity and last is not None:
        try:
            if do_gather:
                from dust3r_amd.parallel import unpack_predictions
                p1, p2 = unpack_predictions(last)
            else:
                p1, p2 = last
            full = (p1['pts3d'], p1['conf'], p2['pts3d_in_other_view'], p2['conf'])
            result['parity_check'] = {'batch_vs_single_pair_calls': parity_batch_vs_single(model, v1, v2, full, sorted({0, B // 2 - 1 if B > 1 else 0, B - 1}))}
            log(f"[bench] parity_check (timed batch vs one-pair calls): {result['parity_check']['batch_vs_single_pair_calls']}")
            del p1, p2, full
        except Exception as e:
            result['parity_check'] = {'error': repr(e)}
    last = None

    # ---- live per-kernel timing (HIP events on the launch stream, outside the timed region) ---------------------
    if rank == 0 and not args.no_profile:
        STAGE[0] = 'c2: profiled forward'
        tele2 = Telemetry(device).start()
        blk = profile_mode(model, v1, v2, args.precision)
        t2 = tele2.stop()
        if blk:
            result.update(blk)
            rf = result.get('roofline')
            if rf and telemetry and telemetry.get('sclk_mhz_mean'):
                # the peak in `roofline.peak` is the nominal 2.4 GHz figure; the chip ran the timed steps at sclk_mhz_mean: frac against the peak AT THAT CLOCK
                rf['frac_at_sampled_clock'] = rf['frac'] * 2400.0 / telemetry['sclk_mhz_mean']
                rf['sampled_clock_note'] = ('frac x 2400 MHz / telemetry.sclk_mhz_mean (the clock the power management held during the timed steps'
                                            + (f'; {t2["sclk_mhz_mean"]:.0f} MHz during the profiled single-stream forward' if t2 and t2.get('sclk_mhz_mean') else '') + ')')

    # ---- small batches: the reference's own call shape (dust3r/demo.py:156 batch_size=1, visloc.py:88 one pair per query) ----------------
    if rank == 0 and world == 1 and not args.no_latency:
        try:
            lat = {}
            for nb, reps in ((1, 20), (2, 12), (4, 8), (8, 5)):
                sub = lambda v: {k: (x[:nb] if isinstance(x, torch.Tensor) else x[:nb]) for k, x in v.items()}   # noqa: E731
                a, b = sub(v1), sub(v2)
                for _ in range(3):
                    model(a, b)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(reps):
                    model(a, b)
                torch.cuda.synchronize()
                ms = (time.perf_counter() - t1) / reps * 1e3
                lat[f'pairs_{nb}'] = {'ms_per_call': ms, 'pairs_per_s': nb / ms * 1e3}
            result['latency'] = dict(lat, what='engine forward, inputs and outputs resident in HBM, back-to-back calls of 1 / 2 / 4 / 8 pairs of 512x384 (same weights and precision as the headline)')
            log('[bench] latency: ' + ', '.join(f"{k} {v['ms_per_call']:.2f} ms" for k, v in lat.items()))
        except Exception as e:
            result['latency'] = {'error': repr(e)}
        # the same calls through the PUBLIC API, host to host: CPU images in, CPU predictions out (what dust3r/demo.py:156 and visloc.py:88 time)
        try:
            from dust3r_amd.image_pairs import make_pairs
            from dust3r_amd.inference import inference
            from dust3r_amd.synthetic import synthetic_image_list
            host = {}
            imgs = synthetic_image_list(100, H, W, seed=0)
            one = [(imgs[0], imgs[1])]
            for _ in range(3):
                inference(one, model, device, batch_size=1, verbose=False)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(20):
                inference(one, model, device, batch_size=1, verbose=False)
            torch.cuda.synchronize()
            host['one_pair_ms'] = (time.perf_counter() - t1) / 20 * 1e3
            many = make_pairs(imgs, scene_graph='swin-3', prefilter=None, symmetrize=True)
            inference(many[:64], model, device, batch_size=32, verbose=False)
            runs = []
            for _ in range(2):
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                out = inference(many, model, device, batch_size=32, verbose=False)
                torch.cuda.synchronize()
                runs.append(time.perf_counter() - t1)
                del out
            host['pairs_600_s'], host['pairs_600_pairs_per_s'] = min(runs), len(many) / min(runs)
            host['what'] = ('dust3r_amd.inference.inference() host to host (CPU image tensors in, CPU prediction + view tensors out, the reference\'s return format): one pair per call '
                            '(batch_size=1), and BASELINE configs[4]\'s pair list (100 views, swin-3 symmetrised = 600 pairs, each distinct image encoded once), best of two')
            result['latency']['public_api_host_to_host'] = host
            log(f"[bench] public API host to host: one pair {host['one_pair_ms']:.2f} ms, 600 pairs {host['pairs_600_s']:.2f} s = {host['pairs_600_pairs_per_s']:.0f} pairs/s")
            del imgs, many, one
        except Exception as e:
            if isinstance(result.get('latency'), dict):
                result['latency']['public_api_host_to_host'] = {'error': repr(e)}

    # ---- opt-in fast modes: NOT parity-grade, reported with their measured error and their own roofline block --------------
    # Error = per-pixel relative pointmap difference against the headline (parity-grade) engine on the same weights and the
    # same first two pairs of the batch; the headline mode itself is held to 1e-3 against the CPU oracle by tests/test_forward_gpu.py.
    if rank == 0 and world == 1 and not args.no_fast:
        fast = {}
        try:
            sub = lambda v, n: {k: (x[:n] if isinstance(x, torch.Tensor) else x[:n]) for k, x in v.items()}   # noqa: E731
            w1, w2 = sub(v1, 2), sub(v2, 2)
            r1, r2 = model(w1, w2)
            ref = torch.cat((r1['pts3d'], r2['pts3d_in_other_view'])).clone()
            for prec in [p for p in ('fp16x2f8', 'fp16f8', 'fp16x3', 'bf16', 'fp16') if p != args.precision]:
                model.set_precision(prec)
                if args.single_stream:
                    model.set_two_streams(False)
                e1, e2 = model(w1, w2)
                got = torch.cat((e1['pts3d'], e2['pts3d_in_other_view']))
                rel = ((got - ref).norm(dim=-1) / ref.norm(dim=-1).clamp_min(1e-8)).flatten()
                for _ in range(2):
                    model(v1, v2)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                nrep = 5
                for _ in range(nrep):
                    model(v1, v2)
                torch.cuda.synchronize()
                dtf = (time.perf_counter() - t1) / nrep
                fast[prec] = {'value': B / dtf, 'unit': 'pairs/s', 'ms_per_step': dtf * 1e3,
                              'frac_of_bf16_mfma_peak': B / dtf * GFLOP_PER_PAIR / 1e3 / PEAK_BF16_TFLOPS,
                              'rel_pointmap_err_vs_headline': {'max': float(rel.max()), 'p99.99': float(rel.kthvalue(int(0.9999 * rel.numel())).values),
                                                               'p99': float(rel.kthvalue(int(0.99 * rel.numel())).values), 'mean': float(rel.mean())},
                              'parity': 'parity-grade (22-bit operands everywhere: max 7e-5 vs the CPU oracle on the full-size model)' if prec == 'fp16x3'
                              else 'opt-in; held to the default mode\'s assertions against the CPU oracle (six weight seeds: per-pixel max <= 5.1e-4, p99.99 <= 1.3e-4, mean <= 1.8e-5: '
                                   'tests/test_timed_configs_gpu.py) -- 22-bit weights, ~15-bit activations, 2.5 MFMA units per product in the transformer blocks\' linears' if prec == 'fp16x2f8'
                              else 'opt-in: NOT claimed to meet the 1e-3 per-pixel bar'}
                if not args.no_profile and prec in ('fp16x2f8', 'fp16f8', 'fp16x3'):
                    blk = profile_mode(model, v1, v2, prec, quiet=True)
                    if blk:
                        fast[prec]['roofline'] = blk['roofline']
                log(f"[bench] mode {prec}: {B / dtf:.1f} pairs/s, rel err vs headline max {float(rel.max()):.2e} mean {float(rel.mean()):.2e}")
            model.set_precision(args.precision)
            if args.single_stream:
                model.set_two_streams(False)
        except Exception as e:
            fast['error'] = repr(e)
        result['fast_mode'] = fast

    if world > 1:
        dist.barrier()
    # ---- second half of the metric + CPU baselines: rank 0 at N = 1 only ----------------------------------------
    if rank == 0 and world == 1:
        del v1, v2
        scene_io = None
        if not args.no_aligner:
            try:
                result['aligner'], scene_io = bench_aligner(device)
                log(f"[bench] aligner {result['aligner']['value']:.1f} iters/s, {result['aligner']['roofline']['achieved']:.0f} GB/s algorithmic")
                try:        # the same 20 views with the demo's symmetrised pair graph (380 edges, 2.48 GB per iteration): reported beside the BASELINE scene
                    if args.no_aligner_380:
                        raise RuntimeError('skipped (--no-aligner-380)')
                    a380, _ = bench_aligner(device, symmetrize=True)
                    result['aligner']['symmetrized_380_edges'] = {k: a380[k] for k in ('value', 'unit', 'n_edges', 'ms_total', 'ms_runs', 'final_loss', 'roofline')}
                    log(f"[bench] aligner, 380 edges: {a380['value']:.1f} iters/s, {a380['roofline']['achieved']:.0f} GB/s algorithmic")
                    del a380, _
                except Exception as e:
                    result['aligner']['symmetrized_380_edges'] = {'error': repr(e)}
            except Exception as e:  # keep the forward line even if the second leg fails
                result['aligner'] = {'error': repr(e)}
        if not args.no_cpu_baseline:
            try:
                keep = {}
                result['cpu_baseline'] = cpu_baseline_forward(keep=keep)
                result['cpu_baseline']['gpu_over_cpu'] = result['value'] / result['cpu_baseline']['value']
                if not args.no_parity and keep:
                    try:        # the engine with the oracle's weights vs the oracle's own output (this overwrites the bench weights: every timed leg is done)
                        pc = parity_vs_cpu_oracle(model, keep['oracle'], *keep['views'], keep['ref'])
                        result.setdefault('parity_check', {})['vs_cpu_oracle'] = pc
                        log(f'[bench] parity_check (engine vs CPU oracle): {pc}')
                    except Exception as e:
                        result.setdefault('parity_check', {})['vs_cpu_oracle'] = {'error': repr(e)}
                    keep.clear()
                if scene_io is not None:
                    cb = cpu_baseline_aligner(scene_io)
                    cb['gpu_over_cpu'] = result['aligner']['value'] / cb['value']
                    result['aligner']['cpu_baseline'] = cb
            except Exception as e:
                result['cpu_baseline'] = {'error': repr(e)}
    if rank == 0:
        emit(result)
    if world > 1:
        dist.barrier()
    if dist.is_initialized():
        dist.destroy_process_group()


def validate_checkpoint_first(path):
    """tools/validate_checkpoint.py on `path` in a subprocess (one synthetic 512x384 pair, default and fp32 engines against the CPU oracle): its verdict as a dict."""
    import subprocess
    if not os.path.exists(path):
        return {'checkpoint': path, 'error': 'DUST3R_CKPT does not exist'}
    tool = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tools', 'validate_checkpoint.py')
    STAGE[0] = 'validate checkpoint'
    t0 = time.perf_counter()
    try:
        r = subprocess.run([sys.executable, tool, path, '--pairs', '1'], capture_output=True, text=True, timeout=int(os.environ.get('D3R_BENCH_CKPT_TIMEOUT', '900')))
        out = {'checkpoint': path, 'exit_code': r.returncode, 'pass': r.returncode == 0, 'seconds': time.perf_counter() - t0,
               'what': 'tools/validate_checkpoint.py <ckpt> --pairs 1: exit 0 = state dict matches the restated module tree key for key AND per-pixel max <= 1e-3 for fp16x3 and fp32',
               'tail': (r.stdout + r.stderr)[-1500:]}
    except Exception as e:          # a timeout or a missing interpreter must not cost the bench line
        out = {'checkpoint': path, 'error': repr(e), 'pass': False}
    log(f'[bench] checkpoint validation: {out}')
    return out


def guarded_main():
    """Scale-run hygiene (round 6): a rank that fails -- RCCL initialisation, the all-gather, an engine error -- or exceeds its wall-clock limit prints ONE JSON line
    {"error": ..., "rank": ..., "stage": ...} on stdout and exits non-zero, instead of a hang or a bare traceback: the first 8-GPU run is diagnosable from its output.
    D3R_BENCH_RANK_TIMEOUT (seconds, default 1500) bounds every rank; collectives carry the process group's own timeout (D3R_BENCH_COLLECTIVE_TIMEOUT, default 300 s)."""
    import signal
    import traceback
    rank = int(os.environ.get('RANK', '0'))

    def fail(kind, detail):
        tail = nccl_log_tail(rank)
        print(json.dumps({'error': kind, 'detail': detail[-2000:], 'rank': rank, 'world_size': int(os.environ.get('WORLD_SIZE', '1')), 'stage': STAGE[0],
                          'host': os.uname().nodename, 'pid': os.getpid(), 'nccl_log_tail': tail}), flush=True)

    def on_alarm(signum, frame):
        fail('rank timeout', f'rank {rank} exceeded D3R_BENCH_RANK_TIMEOUT in stage {STAGE[0]!r}:\n' + ''.join(traceback.format_stack(frame)[-6:]))
        os._exit(3)
    if 'WORLD_SIZE' in os.environ or '--gpus' not in sys.argv:
        signal.signal(signal.SIGALRM, on_alarm)
        signal.alarm(int(os.environ.get('D3R_BENCH_RANK_TIMEOUT', '1500')))
    try:
        main()
    except SystemExit:
        raise
    except BaseException as e:      # noqa: BLE001 -- whatever it is, it becomes the one line
        fail(type(e).__name__, traceback.format_exc())
        os._exit(2)


if __name__ == '__main__':
    guarded_main()
