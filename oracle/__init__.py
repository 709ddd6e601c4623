"""CPU oracle for the GraphCast encode-process-decode path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``graphcast_amd/`` imports this
package; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` do, and only as the checker (never as the thing measured
or shipped).

The oracle is a numpy restatement of the reference algorithm
(``/root/reference/weathernext``), one module per reference file:

===========================  ==================================================
oracle module                reference file it follows
===========================  ==================================================
``mesh.py``                  ``utils/icosahedral_mesh.py:79-133,136-263,321-388``
``connectivity.py``          ``utils/legacy/grid_mesh_connectivity.py:22-134``
``features.py``              ``utils/model_utils.py:29-152,180-216,237-642``
``stacking.py``              ``utils/model_utils.py:155-177,645-776``
``gnn.py``                   ``utils/legacy/deep_typed_graph_net.py:180-401``,
                             ``utils/typed_graph_net.py:272-350,369-546,590-696``
``params.py``                haiku naming applied to ``deep_typed_graph_net.py:205-323``
``graphcast.py``             ``weathernext1_graph/graphcast.py:184-292,298-329,380-730``
===========================  ==================================================

Pinning status
--------------
* STRUCTURE (mesh hierarchy, multi-mesh edges, radius query, structural
  node/edge features, lat/lon->xyz): PINNED.  ``tests/golden/make_golden.py``
  executes the reference's own modules in this container (numpy/scipy only,
  with ``jax``/``xarray``/``trimesh`` stubbed in ``sys.modules``) and commits
  fingerprints + small arrays under ``tests/golden/``; the reference's own
  known-answer tests (``icosahedral_mesh_test.py:72-91``,
  ``grid_mesh_connectivity_test.py:23-47``) are restated in ``tests/``.
* WIRING of the GNN (which MLP sees which concatenation, residuals, parameter
  names): PINNED by executing the reference's ``deep_typed_graph_net.py`` /
  ``typed_graph_net.py`` / ``graphcast.py`` source unmodified on top of
  numpy-backed stand-ins for haiku/jraph/jax (``tests/golden/ref_shims``),
  see ``tests/golden/make_golden.py``.
* ARITHMETIC PRIMITIVES (hk.Linear, hk.LayerNorm eps=1e-5, jax.nn.swish,
  jraph.segment_sum): **parity unpinned** -- dm-haiku, jraph and jax are
  un-vendored, un-pinned dependencies (reference ``setup.py:37,42,43``) that
  are not installable here; their published algorithms are restated in
  ``gnn.py`` and cross-checked against torch's independent implementations.
* HOST PLUMBING (Dataset <-> channels stacking, rollout window, normalisation wrapper):
  PINNED by executing the reference's ``rollout.py`` / ``normalization.py`` /
  ``xarray_tree.py`` / ``model_utils.py`` unmodified (``tests/golden/make_golden_rollout.py``
  -> ``tests/golden/rollout_ref.npz``); ``oracle/stacking.py`` is an additional
  container-free restatement.
* mesh2grid containing-triangle query: the reference calls
  ``trimesh.nearest.on_surface`` (trimesh absent, un-pinned,
  ``setup.py:48``); restated from its published algorithm in
  ``connectivity.py`` -- **parity unpinned** for the 254 (0.25 deg) / 63
  (1 deg) grid points that lie exactly on a mesh edge.
"""
