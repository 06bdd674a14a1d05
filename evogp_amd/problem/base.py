class BaseProblem:
    """Interface of a problem (src/evogp/problem/base.py:1-11)."""

    def evaluate(self, forest):
        raise NotImplementedError

    @property
    def problem_dim(self):
        raise NotImplementedError

    @property
    def solution_dim(self):
        raise NotImplementedError
