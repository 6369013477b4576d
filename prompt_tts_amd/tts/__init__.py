"""Drop-in mirror of the reference's `tts` package surface for the hot path (models, dataloader)."""
