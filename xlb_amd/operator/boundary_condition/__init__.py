from .boundary_condition import (
    BoundaryCondition as BoundaryCondition,
    ImplementationStep as ImplementationStep,
    boundary_condition_registry as boundary_condition_registry,
    EquilibriumBC as EquilibriumBC,
    HalfwayBounceBackBC as HalfwayBounceBackBC,
    FullwayBounceBackBC as FullwayBounceBackBC,
    DoNothingBC as DoNothingBC,
    ZouHeBC as ZouHeBC,
    RegularizedBC as RegularizedBC,
    ExtrapolationOutflowBC as ExtrapolationOutflowBC,
    HybridBC as HybridBC,
)
