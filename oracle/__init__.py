"""CPU oracle for the MAML/ANIL inner/outer-loop hot path.  TEST INFRASTRUCTURE ONLY.

This package restates, on the CPU, the algorithm of the reference path (Kostis-S-Z/exploring_meta:
``core_functions/vision.py``, ``core_functions/vision_models.py``, ``utils/data_pre.py::prepare_batch``,
``vision/maml_vision.py`` outer accumulation, ``core_functions/policies.py``, TRPO part of
``core_functions/rl.py``).  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import it, and only as the checker / reported baseline -- never as the product path.  Nothing under
``exploring_meta_amd/`` imports it; the product fails loudly when its HIP library is missing.

Pinning status
--------------
* vision path (``vision_ref.py``): PINNED against the reference's own code run in the build container:
  ``tests/golden/make_golden.py`` imports ``/root/reference`` (with inert stubs for the absent third-party packages) and
  records outputs of the reference's ``prepare_batch``, ``accuracy``, ``fast_adapt``, ``MiniImagenetCNN``,
  ``OmniglotCNN``, ``ConvBase``, ``DiagNormalPolicy*``; ``tests/test_oracle_golden.py`` checks this oracle against them.
* The learn2learn pieces the reference calls but does not vendor (``MAML.clone/adapt``, ``clone_module``, ``maml_update``;
  version unpinned, not in requirements.txt) are restated from their published semantics (reference
  ``vision/README.md:59-80``, call sites ``core_functions/rl.py:368-374``): PARITY UNPINNED at that boundary.
* RL path (``rl_ref.py``): COMPOSITION PINNED, cherry / learn2learn LEAVES UNPINNED.  ``tests/golden/make_golden_rl.py`` imports the
  reference's ``core_functions/rl.py`` (same stub recipe) and EXECUTES its ``compute_advantages`` (:95-110), ``trpo_a2c_loss``
  (:346-358), ``trpo_update`` (:361-374), ``fast_adapt_trpo`` (:377-406), ``meta_surrogate_loss`` (:441-473) and
  ``meta_optimize_trpo`` (:409-438) on seeded replays (MAML-TRPO small / two inner steps / ANIL-TRPO / BASELINE config 5 at full
  size), with ``rl_ref.py``'s leaf restatements installed where the reference calls cherry / learn2learn; ``tests/test_oracle_rl.py``
  holds ``rl_ref.py``'s own composition to those records at 1e-9.  The leaves themselves (``td.discount``,
  ``pg.generalized_advantage``, ``normalize``, ``LinearValue``, ``a2c/trpo.policy_loss``, ``trpo.hessian_vector_product``,
  ``trpo.conjugate_gradient``, ``clone_module``, ``maml_update``; unpinned in ``requirements.txt:6`` / absent from it) are restated
  from published semantics: PARITY UNPINNED at the leaves.
"""
