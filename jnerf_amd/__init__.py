"""jnerf_amd — the MI355X (gfx950) Instant-NGP hot path behind JNeRF's encoder / sampler / network module API."""
__version__ = "0.1.0"
