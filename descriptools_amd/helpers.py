"""Host logic mirror of descriptools/helpers.py."""
import math

import numpy as np


def divisor(row_length, column_length, row_division, column_division):
    """Tile boundaries floor((i+1)*L/(div+1)), i < div (helpers.py:5-18)."""
    boundary_row = np.array([math.floor((i + 1) * row_length / (row_division + 1))
                             for i in range(row_division)], dtype=int)
    boundary_column = np.array([math.floor((i + 1) * column_length / (column_division + 1))
                                for i in range(column_division)], dtype=int)
    return boundary_row, boundary_column
