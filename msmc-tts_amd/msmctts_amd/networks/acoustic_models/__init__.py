from .multi_stage_predictor import MultiStagePredictor  # noqa: F401
