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
* cherry-rl pieces (``td.discount``, ``pg.generalized_advantage``, ``normalize``, ``LinearValue``, TRPO helpers; unpinned in
  ``requirements.txt:6``) likewise restated from published semantics: PARITY UNPINNED (``rl_ref.py``).
"""
