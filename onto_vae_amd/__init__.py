"""MI355X-native OntoVAE hot path (drop-in for onto_vae.vae_model.OntoVAE)."""
