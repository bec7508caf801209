from .user_api import user_dirichletBC, user_dirichletBC_values

__all__ = ["user_dirichletBC", "user_dirichletBC_values"]
