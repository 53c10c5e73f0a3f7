"""Drivers on either side of the hot path: greedy decision-transformer rollout, MCTS, CLI."""
