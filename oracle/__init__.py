"""CPU oracle for the RQ-VAE + RQ-Transformer sampling path.

TEST INFRASTRUCTURE ONLY.  This package is a plain-numpy restatement of the
reference algorithm (kakaobrain/rq-vae-transformer, files cited per function).
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and only as the checker -- never as the thing shipped or
measured.  The product path (``rq-vae-transformer_amd/``) never imports this
package and fails loudly when the HIP library is missing.

Pinning: the reference ships no tests or golden vectors, so the
oracle is pinned against outputs of the reference itself, imported on CPU in the
build container by ``tests/golden/make_golden.py`` (fixtures committed under
``tests/golden/``; ``tests/test_oracle_golden.py`` re-checks them on every run).

``oracle/_ref/`` (git-ignored build output of ``oracle/build_ref.py``) holds the reference's own modules and its
throughput driver as sourceless bytecode; it is used only by ``bench.py``'s ``cpu_baseline`` leg (through
``oracle/ref_cpu_baseline.py``, a process of its own) and by ``tests/test_gpu_reference_driver.py`` /
``tests/test_oracle_ref.py``.  Importing this package never touches it.
"""
from . import backend  # noqa: F401
from .weights import make_params, rqvae_param_shapes, rqt_param_shapes  # noqa: F401
from .rq import (compute_distances, rq_quantize, rq_embed_code,  # noqa: F401
                 rq_embed_code_with_depth, rq_quantize_margins, rq_soft_codes, vq_ema_step, rq_quantize_train, ema_restart_candidates)
from .sampler import top_k_logits, top_p_probs, filtered_probs  # noqa: F401
from .transformer import RQTransformerOracle  # noqa: F401
from .vae import RQVAEOracle  # noqa: F401
