"""CPU oracle for the MNK self-play rollout path -- TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement of the algorithm in the reference's
``src/env/torch_vector_mnk_env.py`` and ``src/selfplay/torch_self_play_wrapper.py``
(plus the masked-sampling step of ``src/selfplay/policy.py``).  It exists so that
the HIP kernels can be checked bit-for-bit against something that runs without
a GPU and without the reference tree.

Who may import it: ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` -- as the checker / the timed CPU baseline,
never as the product.  Nothing under ``rl-selfplay-mnk_amd/`` imports this
package; the product path raises if the HIP library is missing.

Parity status: PINNED.  ``oracle/pin_against_reference.py`` (run in the build
container, where ``/root/reference`` is mounted) imports the reference and
checks this restatement against it after every operation on randomized traces,
and ``tests/golden/*.npz`` (made by ``tests/golden/make_golden.py`` from the
imported reference) pin it wherever the reference tree is absent.

Modules
-------
env_torch        OracleVectorEnv   -- restates TorchVectorMnkEnv (torch eager, conv2d win scan)
selfplay_torch   OracleSelfPlay    -- restates TorchSelfPlayWrapper
policies         row-local deterministic policies + OracleRandomPolicy
packing          numpy bit-plane pack / unpack (guard-column layout used by the HIP state)
philox           numpy Philox4x32-10 and the uniform-legal-cell sampler of the HIP kernels
rollout          the fused random rollout, agent-step and GAE restated on top of the above
"""
