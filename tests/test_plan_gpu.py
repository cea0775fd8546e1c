"""The launch-plan machinery tests of tests/test_plan.py on the real library (`-m gpu`): recording outside stream
capture executes and logs, replays re-issue on the recorded stream; plus the rebindable-input search, which needs real
argument blocks.  (Trainer-level: tests/test_graph_gpu.py.)"""
import pytest

import test_plan as T

pytestmark = pytest.mark.gpu


def test_plan_replays_recorded_launches_on_current_buffer_contents_gpu():
    T.test_plan_replays_recorded_launches_on_current_buffer_contents()


def test_plan_recording_is_exclusive_gpu():
    T.test_plan_recording_is_exclusive_and_replay_needs_a_finished_plan()


def test_plan_hand_offs_gpu():
    T.test_plan_hand_offs_are_logged_in_call_order()


def test_plan_input_rebinding_gpu():
    T.test_plan_input_can_be_rebound_to_another_buffer()
