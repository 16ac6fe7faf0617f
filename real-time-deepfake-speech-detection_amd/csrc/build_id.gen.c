const char afx_build_id_str[] = "90286387882f";
