"""Host logic of the trainer on the CPU (emulated C ABI) against the golden iteration."""
import step_checks


def test_step_matches_reference(emu_backend):
    step_checks.check_step('cpu')
