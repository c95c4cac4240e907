"""Outer-step optimisers (reference: meta_policy_search/optimizers): maml_first_order_optimizer, conjugate_gradient_optimizer."""
