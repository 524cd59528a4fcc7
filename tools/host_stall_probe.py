"""One pair through inference()'s stages with a timer around each, six times, then with torch on one thread: shows the sporadic 30-100 ms stalls of
host-side tensor operations when torch's intra-op pool (128 threads on the MI355X boxes: 256 logical CPUs) exceeds the container's CPU quota (16), and
their absence once dust3r_amd caps the pool (utils/device.py:fit_host_threads). Usage: python tools/host_stall_probe.py"""
import sys, time, torch
sys.path.insert(0, '.')
import bench
from dust3r_amd import inference as I
from dust3r_amd.utils.device import collate_with_cat
from dust3r_amd.synthetic import synthetic_image_list
dev = torch.device('cuda:0')
model = bench.build_model('fp16x3', dev)
imgs = synthetic_image_list(2, bench.H, bench.W, seed=0)
one = [(imgs[0], imgs[1])]
print('threads', torch.get_num_threads(), torch.get_num_interop_threads())
def T():
    return time.perf_counter()
for rep in range(6):
    t0 = T(); b = collate_with_cat(one[0:1]); t1 = T()
    res = I.loss_of_one_batch(b, model, None, dev); t2 = T()
    sink = I._PredictionSink(1, bench.H, bench.W, 'cpu', dev); t3 = T()
    sink.put(0, 1, res['pred1'], res['pred2']); t4 = T()
    sink.finish(); t5 = T()
    v = collate_with_cat(list(one)); t6 = T()
    x = imgs[0]['img']
    c1 = torch.cat([x]); t7 = T()
    c2 = x.clone(); t8 = T()
    c3 = torch.empty_like(x); c3.copy_(x); t9 = T()
    print(f'rep {rep}: collate {1e3*(t1-t0):.2f} | forward enqueue {1e3*(t2-t1):.2f} | sink init {1e3*(t3-t2):.2f} | put {1e3*(t4-t3):.2f} | finish {1e3*(t5-t4):.2f} | collate again {1e3*(t6-t5):.2f} | cat1 {1e3*(t7-t6):.2f} | clone {1e3*(t8-t7):.2f} | empty+copy {1e3*(t9-t8):.2f}')
torch.set_num_threads(1)
for rep in range(3):
    b = collate_with_cat(one[0:1]); res = I.loss_of_one_batch(b, model, None, dev)
    sink = I._PredictionSink(1, bench.H, bench.W, 'cpu', dev); sink.put(0, 1, res['pred1'], res['pred2']); sink.finish(); t5 = T()
    v = collate_with_cat(list(one)); t6 = T()
    print(f'1 thread rep {rep}: collate again {1e3*(t6-t5):.2f}')
