"""CPU oracle for the equiadapt canonicalization hot path.  TEST INFRASTRUCTURE ONLY.

This package restates, on plain torch-CPU ops and in the reference's own *unfused* op order, the
algorithm of the reference hot path.  Every function cites the
reference file:line it follows (paths relative to /root/reference).

Who may import it: ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` -- as the checker / the timed CPU baseline, never as the product.  Nothing under
``equiadapt_amd/`` imports it, and the product path raises when the HIP library is missing.

PARITY PINNING STATUS
---------------------
* torch-only reference pieces (``equiadapt/common/utils.py``, ``equiadapt/common/basecanonicalization.py``,
  ``equiadapt/pointcloud/canonicalization_networks/vector_neuron_layers.py``, and -- with an
  annotation-only ``omegaconf`` stand-in -- ``pointcloud/canonicalization_networks/equivariant_networks.py``
  and ``pointcloud/canonicalization/continuous_group.py``) import unmodified in the build container.
  ``tests/golden/make_golden.py`` ran them and committed their inputs/outputs under ``tests/golden``;
  ``tests/test_oracle_golden.py`` checks this oracle against those vectors.  -> PINNED.
* the image path (``equiadapt/images/**``) needs kornia 0.7.0 / torchvision 0.17.0 / e2cnn, none of
  which exist in the build container (no network).  The functions in ``image_ops.py`` named ``kornia_*``
  and ``tv_*`` restate those third-party algorithms from their published source as recalled, on the torch
  primitives they delegate to (``F.affine_grid``, ``F.grid_sample``, ``F.pad``, ``F.interpolate``).
  No reference-generated vector exists for them and the reference's own image tests assert no values
  (tests/images/canonicalization/test_discrete_group.py ends without an assert).
  THE ONE COMMAND THAT PINS THEM: ``python tests/golden/regen_with_reference.py --reference <unmodified equiadapt checkout>``
  on a machine with kornia 0.7.0 + torchvision 0.17.0 (+ e2cnn): it regenerates every image case through the reference's own
  classes, diffs them against this oracle, and writes ``tests/golden/images_reference.pt``, which
  ``tests/test_oracle_golden.py::test_oracle_matches_reference_generated_image_vectors`` then checks in every CPU run.
  -> **PARITY UNPINNED** at the kornia/torchvision boundary until that file exists; guarded by property tests
  (C4 == torch.rot90, group composition, round trips, pad+rotate+crop == clamp-gather) and by
  restatement-generated fixtures that are labelled as such.
"""
