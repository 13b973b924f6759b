"""Action-noise explorations for the off-policy agents (SAC / TD3) — added with those agents."""
