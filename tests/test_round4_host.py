"""CPU-side checks added in round 4 (ADVICE r3 items + bench launcher + host logic). No GPU needed."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_adam_fallback_with_closure_takes_the_step_and_returns_the_loss():
    from dirhip.optim import Adam
    p = torch.nn.Parameter(torch.randn(5, 3))                     # CPU parameter: torch's own step
    q = p.detach().clone().requires_grad_(True)
    def closure_for(t):
        def c():
            t.grad = None
            l = (t * t).sum()
            l.backward()
            return l
        return c
    a, b = Adam([p], lr=1e-2), torch.optim.Adam([q], lr=1e-2)
    la, lb = a.step(closure_for(p)), b.step(closure_for(q))
    assert la is not None and float(la.detach()) == float(lb.detach())
    assert torch.equal(p, q) and not torch.equal(p.detach(), torch.zeros_like(p))
    assert float(a.state[p]["step"]) == 1.0
