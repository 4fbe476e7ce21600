"""Regression bases for Longstaff-Schwartz (reference: maths/regression.py:3-14)."""
import torch


class RegressionFunction:
    def __init__(self, degree):
        self.degree = degree

    def get_degree(self):
        # number of basis functions, as in the reference (degree + 1)
        return self.degree + 1


class PolyomialRegression(RegressionFunction):  # (sic) the reference's spelling is part of its API
    """Monomial basis [x^0 .. x^degree]."""

    def get_regression_matrix(self, explanatory_variables: torch.Tensor) -> torch.Tensor:
        cols = [torch.ones_like(explanatory_variables)]
        for _ in range(self.degree):
            cols.append(cols[-1] * explanatory_variables)
        return torch.stack(cols, dim=1)


PolynomialRegression = PolyomialRegression
