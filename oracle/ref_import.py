"""ORACLE (test infrastructure; BUILD CONTAINER ONLY) -- import the UNMODIFIED reference
files from /root/reference so the restatements in this package can be pinned against them.

What is real and what is restated when the reference is imported this way:
  real      dust3r/{model,inference,patch_embed,image_pairs,post_process,optim_factory}.py,
            dust3r/heads/*, dust3r/utils/*, dust3r/cloud_opt/*      (every line that exists)
  restated  `models.*`  <- oracle/croco_ref/models (croco submodule is an empty directory)
            `roma`      <- oracle/roma_ref.py       (not installed)
            `cv2`, `torchvision`, `trimesh` <- oracle/shims (not installed)
/root/reference does not exist on the GPU box: nothing under tests -m gpu, smoke() or
bench.py calls this module; it is used by oracle/make_golden.py and by `-m "not gpu"`
tests that skip when the directory is missing.
"""
import os
import sys
import types

REFERENCE_ROOT = '/root/reference'
_HERE = os.path.dirname(os.path.abspath(__file__))


def reference_available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, 'dust3r'))


def import_reference():
    """Returns the reference's top-level `dust3r` package (imported from /root/reference)."""
    if not reference_available():
        raise ImportError('/root/reference is not present (expected on the GPU box)')
    repo_root = os.path.dirname(_HERE)
    for p in (os.path.join(_HERE, 'shims'), os.path.join(_HERE, 'croco_ref'), REFERENCE_ROOT, repo_root):
        if p not in sys.path:
            sys.path.insert(0, p)
    # dust3r/utils/path_to_croco.py:13-19 raises when <reference>/croco/models is missing;
    # pre-seed an empty module in its place (the restated `models` package is already on sys.path)
    if 'dust3r.utils.path_to_croco' not in sys.modules:
        sys.modules['dust3r.utils.path_to_croco'] = types.ModuleType('dust3r.utils.path_to_croco')
    import dust3r  # noqa
    assert os.path.realpath(dust3r.__file__).startswith(REFERENCE_ROOT), dust3r.__file__
    return dust3r
