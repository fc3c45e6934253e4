from .evaluator import AP, Evaluator, average_precision  # noqa: F401
