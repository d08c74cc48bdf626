"""Committed outputs of the CPU oracle's expensive legs (test infrastructure, like oracle/ itself).

The `-m gpu` suite compares the HIP path with the fp32 CPU oracle at SD1.x width.  The oracle costs 2-8 s per UNet sample-forward, so a suite
that re-runs it spends twenty minutes of host time for seconds of GPU time.  Every oracle leg of the GPU tests is therefore a module-level
function decorated with `@oracle_leg(cases=[...])`: a pure function of its (small, hashable) arguments -- all tensors are regenerated from
seeds inside it -- whose result (latents, eps, best indices, losses, word maps, reference-precision floors) is stored once under
`tests/golden/oracle_cache/<module>.<function>[<args>].npz` by `tests/golden/make_oracle_cache.py` and read back by the tests.

  ETAINV_ORACLE=cache (default)  read the committed file; a missing or STALE file (MANIFEST.json: written from another oracle/*.py or leg source) is
                                 computed live (and reported)
  ETAINV_ORACLE=live             ignore the files and run the oracle (the pre-round-4 behaviour; `tests/test_oracle_cache.py` does this for
                                 every leg under @pytest.mark.slow and for a small one in the default CPU suite, so the files cannot drift)
  ETAINV_ORACLE=write            like cache, and a missing file is written after it has been computed (make_oracle_cache.py --force deletes first)

Nothing outside tests/ imports this module."""
import json
import os
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
CACHE_DIR = ROOT / "tests" / "golden" / "oracle_cache"
LEGS = {}                      # "<module>.<function>" -> (function, [case tuples])
_unets = {}


# ------------------------------------------------------------------------------------------------ shared oracle networks (built on first use)
def oracle_unet():
    """the fp32 CPU oracle UNet at SD1.x width (3.4 GB), one per process"""
    if "fp32" not in _unets:
        from oracle.unet import build_unet
        _unets["fp32"] = build_unet(0)
    return _unets["fp32"]


def lowprec_unet(dtype):
    """the oracle with the reference's 16-bit execution emulated (oracle/lowprec.py), one per dtype and process"""
    if dtype not in _unets:
        from oracle.lowprec import LowPrecisionUNet
        from oracle.unet import build_unet
        _unets[dtype] = LowPrecisionUNet(build_unet(0), dtype)
    return _unets[dtype]


def release_networks():
    _unets.clear()


# ------------------------------------------------------------------------------------------------ (de)serialisation of nested results
def _pack(obj, path, arrays):
    if isinstance(obj, torch.Tensor):
        t = obj.detach().cpu()
        tag = str(t.dtype).replace("torch.", "")
        if t.dtype in (torch.bfloat16, torch.float16):
            t = t.float()
        arrays[path] = t.numpy()
        return {"t": path, "dtype": tag}
    if isinstance(obj, np.ndarray):
        arrays[path] = obj
        return {"n": path}
    if isinstance(obj, dict):
        return {"d": {str(k): _pack(v, f"{path}/{k}", arrays) for k, v in obj.items()}}
    if isinstance(obj, (list, tuple)):
        return {"l": [_pack(v, f"{path}/{i}", arrays) for i, v in enumerate(obj)], "tuple": isinstance(obj, tuple)}
    if isinstance(obj, (np.integer,)):
        return {"v": int(obj)}
    if isinstance(obj, (np.floating,)):
        return {"v": float(obj)}
    if obj is None or isinstance(obj, (bool, int, float, str)):
        return {"v": obj}
    raise TypeError(f"oracle leg results hold tensors, arrays, scalars and containers of them; got {type(obj)} at {path}")


def _unpack(node, arrays):
    if "t" in node:
        return torch.from_numpy(np.array(arrays[node["t"]])).to(getattr(torch, node["dtype"]))
    if "n" in node:
        return np.array(arrays[node["n"]])
    if "d" in node:
        return {k: _unpack(v, arrays) for k, v in node["d"].items()}
    if "l" in node:
        seq = [_unpack(v, arrays) for v in node["l"]]
        return tuple(seq) if node["tuple"] else seq
    return node["v"]


def save(path, obj):
    arrays = {}
    skel = _pack(obj, "r", arrays)
    path.parent.mkdir(parents=True, exist_ok=True)
    np.savez_compressed(path, __skeleton__=np.frombuffer(json.dumps(skel).encode(), dtype=np.uint8), **arrays)


def load(path):
    with np.load(path) as z:
        skel = json.loads(bytes(z["__skeleton__"]).decode())
        return _unpack(skel, z)


# ------------------------------------------------------------------------------------------------ fingerprints: a committed result belongs to ONE oracle
# A result file is keyed by leg name and arguments only; what it was computed FROM is recorded next to it: MANIFEST.json maps each key to the
# fingerprint of oracle/*.py and of the leg's own source at the time it was written (comments, docstrings and formatting do not count: the hash is
# taken over the AST).  A mismatch makes the entry STALE: the wrapper recomputes it live (and says so), `make_oracle_cache.py --list` reports it and
# the CPU suite fails (tests/test_oracle_cache.py) -- an edit of the oracle can no longer leave the GPU suite comparing against yesterday's oracle.
MANIFEST = CACHE_DIR / "MANIFEST.json"
_fp_cache = {}


def _ast_hash(src):
    import ast
    import hashlib
    import textwrap
    tree = ast.parse(textwrap.dedent(src))
    for node in ast.walk(tree):                      # docstrings are not behaviour
        if isinstance(node, (ast.FunctionDef, ast.ClassDef, ast.Module, ast.AsyncFunctionDef)) and node.body and \
                isinstance(node.body[0], ast.Expr) and isinstance(getattr(node.body[0], "value", None), ast.Constant) and isinstance(node.body[0].value.value, str):
            node.body = node.body[1:] or [ast.Pass()]
    return hashlib.sha256(ast.dump(tree, include_attributes=False).encode()).hexdigest()[:16]


def oracle_fingerprint():
    if "oracle" not in _fp_cache:
        import hashlib
        h = hashlib.sha256()
        for f in sorted((ROOT / "oracle").glob("*.py")):
            h.update(f.name.encode())
            h.update(_ast_hash(f.read_text()).encode())
        _fp_cache["oracle"] = h.hexdigest()[:16]
    return _fp_cache["oracle"]


def leg_fingerprint(fn):
    """the leg's own source plus, one level deep, the helpers of its module it names (input builders, shared configuration constants)"""
    import ast
    import hashlib
    import inspect
    import textwrap
    src = inspect.getsource(fn)
    h = hashlib.sha256(_ast_hash(src).encode())
    names = sorted({n.id for n in ast.walk(ast.parse(textwrap.dedent(src))) if isinstance(n, ast.Name)})
    for nm in names:
        obj = fn.__globals__.get(nm)
        if inspect.isfunction(obj) and obj.__module__ == fn.__module__ and obj is not fn:
            h.update(nm.encode())
            h.update(_ast_hash(inspect.getsource(getattr(obj, "raw", obj))).encode())
        elif isinstance(obj, (int, float, str, tuple, list, dict)) and nm.isupper():
            h.update(f"{nm}={obj!r}".encode())
    return h.hexdigest()[:16]


def read_manifest():
    return json.loads(MANIFEST.read_text()) if MANIFEST.exists() else {}


def record(key, fn):
    m = read_manifest()
    m[key] = {"oracle": oracle_fingerprint(), "leg": leg_fingerprint(fn)}
    MANIFEST.write_text(json.dumps(dict(sorted(m.items())), indent=1) + "\n")


def stale_reason(key, fn):
    """None if the committed result of `key` was computed from today's oracle/ and leg source (or predates the manifest: no entry), else what changed"""
    e = read_manifest().get(key)
    if e is None:
        return None
    if e["oracle"] != oracle_fingerprint():
        return "oracle/*.py changed since the result was written"
    if e["leg"] != leg_fingerprint(fn):
        return "the leg's source changed since the result was written"
    return None


# ------------------------------------------------------------------------------------------------ the decorator
def _argstr(a):
    if isinstance(a, torch.dtype):
        return {"torch.float16": "fp16", "torch.bfloat16": "bf16", "torch.float32": "fp32", "torch.float64": "fp64"}[str(a)]
    if isinstance(a, dict):
        return "{" + ",".join(f"{k}={_argstr(v)}" for k, v in sorted(a.items())) + "}"
    if isinstance(a, (list, tuple)):
        return "(" + ",".join(_argstr(v) for v in a) + ")"
    if isinstance(a, float):
        return repr(a)
    return str(a)


def key_of(name, args):
    return f"{name}[{','.join(_argstr(a) for a in args)}]" if args else name


def mode():
    m = os.environ.get("ETAINV_ORACLE", "cache")
    assert m in ("cache", "live", "write"), m
    return m


def oracle_leg(cases=((),)):
    """Register `fn(*args)` as an oracle leg.  `cases` lists the argument tuples the tests use (what make_oracle_cache.py generates)."""
    def deco(fn):
        name = f"{fn.__module__.rsplit('.', 1)[-1]}.{fn.__name__}"
        LEGS[name] = (fn, [tuple(c) for c in cases])
        memo = {}

        def wrapper(*args):
            key = key_of(name, args)
            if key in memo:
                return memo[key]
            path = CACHE_DIR / f"{key}.npz"
            m = mode()
            why = stale_reason(key, fn) if (m != "live" and path.exists()) else None
            if m != "live" and path.exists() and why is None:
                res = load(path)
            else:
                if m == "cache":
                    print(f"[oracle_cache] {key}: {'STALE (' + why + ')' if why else 'no committed result'}, running the oracle live", file=sys.stderr, flush=True)
                with torch.no_grad():
                    res = fn(*args)
                if m == "write":
                    save(path, res)
                    record(key, fn)
                    res = load(path)                     # what the tests will see
            memo[key] = res
            return res
        wrapper.__name__, wrapper.__doc__, wrapper.leg_name, wrapper.raw = fn.__name__, fn.__doc__, name, fn
        return wrapper
    return deco
