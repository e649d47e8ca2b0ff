"""MI355X-native OntoVAE hot path (drop-in for onto_vae.vae_model.OntoVAE)."""


def __getattr__(name):
    """onto_vae_amd.OntoVAE / onto_vae_amd.VAE / onto_vae_amd.OntoEncVAE without importing torch until they are asked for."""
    if name in ("OntoVAE", "VAE"):
        from . import vae_model
        return getattr(vae_model, name)
    if name == "OntoEncVAE":
        from . import enc_vae_model
        return enc_vae_model.OntoEncVAE
    raise AttributeError(name)
