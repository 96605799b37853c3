"""Drop-in mirror of the reference's `src` package for the stage-2 path (muzishen/RCDMs):
`src.models.*` and `src.pipelines.RCDMs_pipeline` keep the reference's class names, constructor
kwargs, state-dict keys and call signatures; the compute underneath is rcdms_amd (HIP, gfx950)."""
