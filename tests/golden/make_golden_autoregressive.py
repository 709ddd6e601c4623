"""tests/golden/autoregressive_ref.npz: the REFERENCE's weathernext/utils/autoregressive.py
(``Predictor.__call__`` :127-222, ``_update_inputs`` :114-125) executed unmodified around the
same toy one-step predictor as make_golden_rollout.py.

Stand-ins supplied here (test infrastructure only): ``hk.scan`` as a python loop that stacks the
per-step outputs; ``jax.tree_util`` flatten / unflatten of a Dataset as its variables' arrays in
sorted-name order (what xarray_jax registers); the few ``xarray.Dataset`` methods this module
uses beyond xarray_lite's surface (transpose, isel(drop=), expand_dims, squeeze,
drop_vars(errors=)), written on top of xarray_lite.

    python tests/golden/make_golden_autoregressive.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden_rollout as base                           # noqa: E402  (installs the stand-ins)

xl = base.xarray_lite
import haiku as hk                                           # noqa: E402
import jax                                                   # noqa: E402


# ---- Dataset surface used by autoregressive.py --------------------------------------------
def _ds_transpose(self, *dims):
  return xl.Dataset._construct({k: (v.transpose(*dims) if "time" in v.dims else v)
                                for k, v in self._vars.items()}, self._coords)


def _ds_isel(self, indexers=None, drop=False, **kw):
  indexers = dict(indexers or {}, **kw)
  out = _orig_isel(self, indexers)
  if drop:
    scalar = [d for d, i in indexers.items() if isinstance(i, (int, np.integer))]
    coords = {k: v for k, v in out._coords.items() if not (v.dims == () and k in scalar)}
    out = xl.Dataset._construct(out._vars, coords)
  return out


def _ds_expand_dims(self, dim=None, axis=0, **kw):
  (name, coord), = dict(dim or {}, **kw).items()
  coord = np.asarray(getattr(coord, "values", coord))
  new = {k: xl.Variable((name,) + v.dims, np.broadcast_to(v.data[None], (len(coord),) + v.shape))
         for k, v in self._vars.items()}
  coords = dict(self._coords)
  coords[name] = xl.Variable((name,), coord)
  return xl.Dataset._construct(new, coords)


def _ds_squeeze(self, dim, drop=False):
  new = {k: (v.isel({dim: 0}) if dim in v.dims else v) for k, v in self._vars.items()}
  coords = {k: v for k, v in self._coords.items() if dim not in v.dims}
  return xl.Dataset._construct(new, coords)


def _ds_drop_vars(self, names, errors="raise"):
  names = [names] if isinstance(names, str) else list(names)
  return _orig_drop(self, [n for n in names if n in self._vars or n in self._coords])


_orig_isel, _orig_drop = xl.Dataset.isel, xl.Dataset.drop_vars
xl.Dataset.transpose = _ds_transpose
xl.Dataset.isel = _ds_isel
xl.Dataset.expand_dims = _ds_expand_dims
xl.Dataset.squeeze = _ds_squeeze
xl.Dataset.drop_vars = _ds_drop_vars

# ---- pytree view of a Dataset (xarray_jax's registration) ---------------------------------
_flat, _unflat = jax.tree_util.tree_flatten, jax.tree_util.tree_unflatten


def tree_flatten(tree):
  if isinstance(tree, xl.Dataset):
    names = sorted(tree.keys())
    return [tree._vars[n].data for n in names], ("dataset", names, [tree._vars[n].dims for n in names],
                                                 dict(tree._coords))
  return _flat(tree)


def tree_unflatten(treedef, leaves):
  if isinstance(treedef, tuple) and treedef and treedef[0] == "dataset":
    _, names, dims, coords = treedef
    return xl.Dataset._construct({n: xl.Variable(d, l) for n, d, l in zip(names, dims, leaves)}, coords)
  return _unflat(treedef, leaves)


jax.tree_util.tree_flatten = tree_flatten
jax.tree_util.tree_unflatten = tree_unflatten
jax.tree_util.tree_leaves = lambda t: tree_flatten(t)[0]


def scan(f, init, xs, length=None):
  carry, outs = init, []
  n = len(xs[0]) if isinstance(xs, (list, tuple)) else len(xs)
  for t in range(n):
    carry, y = f(carry, [x[t] for x in xs])
    outs.append(y)
  return carry, [np.stack([o[i] for o in outs]) for i in range(len(outs[0]))]


hk.scan = scan

sys.modules["xarray_jax"] = base._Inert("xarray_jax")
from weathernext.utils import autoregressive as ref_ar       # noqa: E402


def main():
  synthetic, gc = base.synthetic, base.gc
  inputs, template, forcings = synthetic.make_example(base.TASK, base.LAT, base.LON,
                                                      num_target_steps=base.STEPS, seed=base.SEED)
  state = {}

  class RefToy:
    def __call__(self, inputs, targets_template, forcings, **kw):
      mu = base.ref_mu
      x = xl.concat([mu.dataset_to_stacked(inputs), mu.dataset_to_stacked(forcings)], dim="channels")
      data = np.asarray(mu.lat_lon_to_leading_axes(x).data, np.float32)
      if "a" not in state:
        state["a"] = base.toy_weights(data.shape[-1], mu.dataset_to_stacked(targets_template).sizes["channels"])
      y = xl.DataArray(np.tanh(data @ state["a"]), dims=("lat", "lon", "batch", "channels"))
      return mu.stacked_to_dataset(mu.restore_leading_axes(y).variable, targets_template)

  strip = lambda ds: ds.drop_vars(["datetime"])            # autoregressive.py works on relative time only
  preds = ref_ar.Predictor(RefToy())(strip(inputs), strip(template), strip(forcings))
  out = {f"pred:{k}": np.asarray(preds[k].values) for k in preds.keys()}
  out.update({f"dims:{k}": np.array("|".join(preds[k].dims)) for k in preds.keys()})
  out["config"] = np.array([base.STEPS, base.SEED, base.W_SEED])
  path = os.path.join(HERE, "autoregressive_ref.npz")
  np.savez_compressed(path, **out)
  print("wrote", path, {k: v.shape for k, v in list(out.items())[:3]})


if __name__ == "__main__":
  main()
