"""MI355X-native drop-in for the sampling path of kakaobrain/rq-vae-transformer's ``rqvae`` package.

Same import surface as the reference for that path:
``rqvae.models.create_model``, ``rqvae.models.rqvae.RQVAE``, ``rqvae.models.rqtransformer.RQTransformer``,
``rqvae.utils.{config,dist,utils}``.  All arithmetic runs in librqamd.so (hand-written HIP for gfx950,
include/rqamd.h); these modules hold parameters under the reference's state_dict names and marshal
pointers.  There is no CPU fallback."""
