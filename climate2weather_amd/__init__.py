"""climate2weather_amd -- MI355X (gfx950) engine for the Climate2Weather score-based diffusion downscaler hot path."""
from .score import ScoreUNet, timestep_embedding  # noqa: F401

__all__ = ["ScoreUNet", "timestep_embedding"]
