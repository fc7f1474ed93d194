"""Operation / distribution-mode tags (rlaopt/linops/enums.py)."""
from enum import Enum


class _Operation(Enum):
    MATVEC = "matvec"
    RMATVEC = "rmatvec"


class _DistributionMode(Enum):
    ROW = "row"        # operator split by rows: product = concat, transpose = sum
    COLUMN = "column"  # operator split by columns: product = sum, transpose = concat

    @classmethod
    def _from_str(cls, value, param_name):
        if isinstance(value, cls):
            return value
        if isinstance(value, str):
            try:
                return cls(value.lower())
            except ValueError:
                pass
        raise ValueError(
            f"Invalid value for {param_name}: {value}. Expected 'row', 'column', "
            "_DistributionMode.ROW, or _DistributionMode.COLUMN."
        )
