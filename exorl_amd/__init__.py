"""exorl_amd — MI355X (gfx950) backend for the exorl RL-update hot path: HBM-resident replay sampling and
agent.update() for the DDPG-backbone / offline agents, behind the reference's own agent and replay-loader
interfaces. Kernels live in csrc/ (libexorl_hip.so, C ABI in include/exorl_hip.h)."""
