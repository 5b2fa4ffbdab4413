from newtonnet_amd.models.newtonnet import NewtonNet, EmbeddingNet, InteractionNet
from newtonnet_amd.models.output import (CustomOutputSet, DerivativeProperty, DirectProperty, EnergyOutput, DirectForceOutput,
                                         GradientForceOutput, VirialOutput, StressOutput, EnergyAggregator,
                                         NullAggregator, get_output_by_string, get_aggregator_by_string)
