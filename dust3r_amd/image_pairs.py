"""Pair-graph construction -- host-side mirror of the reference `dust3r/image_pairs.py:11-104`
(`make_pairs`, `filter_pairs_seq`, `filter_edges_seq`). The pair list is the unit of data
parallelism: `dust3r_amd.parallel.shard_pairs` gives each rank a contiguous slice of it.

Same names, argument meaning and output ordering as the reference (the windowed graphs go
through a `set` of index pairs exactly like the reference does, because its iteration order is
what fixes the edge order downstream).
"""
import numpy as np
import torch


def _window_size(scene_graph, default=3):
    try:
        return int(scene_graph.split('-')[1])
    except Exception:
        return default


def _index_pairs(n, scene_graph):
    """(i, j) index pairs of the un-symmetrised graph, in reference order."""
    if scene_graph == 'complete':
        return [(i, j) for i in range(n) for j in range(i)]
    if scene_graph.startswith('swin') or scene_graph.startswith('logwin'):
        cyclic = not scene_graph.endswith('noncyclic')
        win = _window_size(scene_graph)
        ids = set()
        if scene_graph.startswith('swin'):
            for i in range(n):
                for off in range(1, win + 1):
                    j = (i + off) % n if cyclic else i + off
                    if j >= n:
                        continue
                    ids.add((i, j) if i < j else (j, i))
        else:
            offsets = [2 ** k for k in range(win)]
            for i in range(n):
                for j in [i - o for o in offsets] + [i + o for o in offsets]:
                    if cyclic:
                        j = j % n
                    if j < 0 or j >= n or j == i:
                        continue
                    ids.add((i, j) if i < j else (j, i))
        return list(ids)
    if scene_graph.startswith('oneref'):
        ref = int(scene_graph.split('-')[1]) if '-' in scene_graph else 0
        return [(ref, j) for j in range(n) if j != ref]
    return []


def make_pairs(imgs, scene_graph='complete', prefilter=None, symmetrize=True):
    pairs = [(imgs[i], imgs[j]) for i, j in _index_pairs(len(imgs), scene_graph)]
    if symmetrize:
        pairs += [(b, a) for a, b in pairs]
    if isinstance(prefilter, str) and prefilter.startswith('seq'):
        pairs = filter_pairs_seq(pairs, int(prefilter[3:]))
    if isinstance(prefilter, str) and prefilter.startswith('cyc'):
        pairs = filter_pairs_seq(pairs, int(prefilter[3:]), cyclic=True)
    return pairs


def sel(x, kept):
    if isinstance(x, dict):
        return {k: sel(v, kept) for k, v in x.items()}
    if isinstance(x, (torch.Tensor, np.ndarray)):
        return x[kept]
    if isinstance(x, (tuple, list)):
        return type(x)([x[k] for k in kept])


def _filter_edges_seq(edges, seq_dis_thr, cyclic=False):
    n = max(max(e) for e in edges) + 1
    kept = []
    for e, (i, j) in enumerate(edges):
        dis = abs(i - j)
        if cyclic:
            dis = min(dis, abs(i + n - j), abs(i - n - j))
        if dis <= seq_dis_thr:
            kept.append(e)
    return kept


def filter_pairs_seq(pairs, seq_dis_thr, cyclic=False):
    edges = [(a['idx'], b['idx']) for a, b in pairs]
    return [pairs[k] for k in _filter_edges_seq(edges, seq_dis_thr, cyclic=cyclic)]


def filter_edges_seq(view1, view2, pred1, pred2, seq_dis_thr, cyclic=False):
    edges = [(int(i), int(j)) for i, j in zip(view1['idx'], view2['idx'])]
    kept = _filter_edges_seq(edges, seq_dis_thr, cyclic=cyclic)
    print(f'>> Filtering edges more than {seq_dis_thr} frames apart: kept {len(kept)}/{len(edges)} edges')
    return sel(view1, kept), sel(view2, kept), sel(pred1, kept), sel(pred2, kept)
