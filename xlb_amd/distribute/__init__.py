from .distribute import (
    init_process_group as init_process_group,
    distribute as distribute,
    gather_field as gather_field,
    SlabPlan as SlabPlan,
    HostStagedHalo as HostStagedHalo,
    all_reduce_sum as all_reduce_sum,
    barrier as barrier,
    all_reduce_max as all_reduce_max,
    all_reduce_min as all_reduce_min,
    all_gather as all_gather,
    shutdown as shutdown,
    transport as transport,
)
