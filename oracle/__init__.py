"""CPU oracle for the flocoder latent-flow hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``flocoder_amd/`` may import this
package: only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg use it, and there only as the checker / the timed CPU
baseline.  The product path is the HIP library and fails loudly without it.
"""
