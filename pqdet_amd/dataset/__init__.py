from .base_sample import RECOVER_BBOXES_REGISTER  # noqa: F401
