"""oracle/ -- CPU restatement of the Adv-GRPO SD3 rollout-and-update hot path.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import anything from this package, and only as the checker /
the reported CPU baseline.  The product path (``adv_grpo_amd``) never imports
it and fails loudly when the HIP extension is missing.

Every function cites the reference file:line it restates (paths relative to
the upstream checkout, e.g. ``adv_grpo/stat_tracking.py:18-47``).

Pinning status (see DESIGN.md "Oracle"):

* pinned against the reference's own importable code via golden vectors made
  by ``tests/golden/make_golden.py`` (run in the build container, where
  /root/reference is mounted): SDE step + log-prob, k-repeat sampler, group
  advantages, zero-std ratio, CLIPCriterion loss, DINO hinge loss,
  multi_score aggregation, GRPO loss + diagnostics, EMA decay,
  dino_patch_cotrain scoring, one full ``pipeline_with_logprob_random``
  trajectory + ``compute_log_prob`` replay on a stand-in transformer.
* pinned against the installed ``transformers`` package (CLIPModel,
  Dinov2Model with seeded random weights): ``oracle.vit``.
* PARITY UNPINNED: ``oracle.scheduler`` (diffusers FlowMatchEuler schedule),
  ``oracle.mmdit`` (diffusers SD3Transformer2DModel) and ``oracle.vae``
  (diffusers AutoencoderKL decoder).  diffusers==0.33.1 is an un-vendored
  dependency that is absent from /root/reference and from this image; those
  three restate its published architecture.
"""
